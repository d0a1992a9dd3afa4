/* Plain-C consumer of librubikhip.so: shows the ABI needs nothing but pointers, sizes and a stream.
 * Built by tests/test_abi_c.py with gcc (no hipcc, no C++, no Python types).  Checks KAT-A of
 * R U R' U' from solved, plus U then U' (reward -1 / +1). */
#define __HIP_PLATFORM_AMD__ 1
#define _POSIX_C_SOURCE 200112L
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rubikhip.h"

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP failure line %d\n", __LINE__); return 2; } } while (0)
#define RC(x) do { if ((x) != RC_OK) { fprintf(stderr, "rc failure line %d: %s\n", __LINE__, rc_last_error()); return 3; } } while (0)

int main(void) {
    const int64_t n = 5, pitch = 256;
    uint8_t *st, *act, *done, host[54 * 256], hdone[8], hact[16];
    float *reward, hrew[8];
    if (rc_version() < 600 || strlen(rc_build_id()) != 16) return 1;      /* the 16 hex digits of the source hash the build embedded */
    /* a caller that skips rc_init gets RC_ENODEV, not a raw launch error */
    if (rc_fill_solved((uint8_t *)host, n, pitch, 3, NULL) != RC_ENODEV || strlen(rc_last_error()) == 0) return 16;
    RC(rc_init(0));
    char what[160];
    RC(rc_describe_dispatch(RC_OP_STEP, 3, (int64_t)1 << 22, 0, RC_OUT_STATES | RC_OUT_DONE, RC_FMT_NONE, 0, what, (int)sizeof what));
    if (strncmp(what, "k_step<Cube3,V=", 15) != 0 || !strstr(what, "POL=1")) { fprintf(stderr, "describe: %s\n", what); return 17; }
    CK(hipMalloc((void **)&st, 54 * pitch)); CK(hipMalloc((void **)&act, 16)); CK(hipMalloc((void **)&done, 16));
    CK(hipMalloc((void **)&reward, 16 * sizeof(float)));
    RC(rc_fill_solved(st, n, pitch, 3, NULL));
    const uint8_t seq[4] = {4, 0, 5, 1};                       /* R U R' U' */
    for (int k = 0; k < 4; ++k) {
        memset(hact, seq[k], sizeof hact);
        CK(hipMemcpy(act, hact, 16, hipMemcpyHostToDevice));
        RC(rc_apply_moves(st, st, act, n, pitch, pitch, 3, reward, done, NULL, RC_FMT_NONE, 0, NULL));
    }
    CK(hipMemcpy(host, st, 54 * pitch, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hrew, reward, sizeof hrew, hipMemcpyDeviceToHost));
    const char *kat = "004002002110511011223220222331333333544444444511555555";
    for (int c = 0; c < n; ++c)
        for (int s = 0; s < 54; ++s)
            if (host[s * pitch + c] != kat[s] - '0') { fprintf(stderr, "KAT-A mismatch cube %d sticker %d\n", c, s); return 4; }
    if (hrew[0] != -1.0f) return 5;
    RC(rc_fill_solved(st, n, pitch, 3, NULL));
    memset(hact, 0, sizeof hact); CK(hipMemcpy(act, hact, 16, hipMemcpyHostToDevice));
    RC(rc_apply_moves(st, st, act, n, pitch, pitch, 3, reward, done, NULL, RC_FMT_NONE, 0, NULL));
    CK(hipMemcpy(hrew, reward, sizeof hrew, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdone, done, 8, hipMemcpyDeviceToHost));
    if (hrew[2] != -1.0f || hdone[2] != 0) return 6;
    memset(hact, 1, sizeof hact); CK(hipMemcpy(act, hact, 16, hipMemcpyHostToDevice));
    RC(rc_apply_moves(st, st, act, n, pitch, pitch, 3, reward, done, NULL, RC_FMT_NONE, 0, NULL));
    CK(hipMemcpy(hrew, reward, sizeof hrew, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdone, done, 8, hipMemcpyDeviceToHost));
    if (hrew[2] != 1.0f || hdone[2] != 1) return 7;
    /* the batch-1 latency path: results arrive in host-mapped pinned memory, no copies, no stream sync */
    uint8_t *pinned;
    CK(hipHostMalloc((void **)&pinned, 8192, hipHostMallocMapped));
    memset(pinned, 0, 8192);
    RC(rc_fill_solved(st, n, pitch, 3, NULL));
    for (int k = 0; k < 4; ++k) RC(rc_facade_step(st, pitch, 3, seq[k], pinned, (uint32_t)(k + 1), 1, NULL));
    const uint8_t kat_cols[20] = {9, 3, 20, 1, 12, 15, 7, 21, 6, 2, 4, 18, 8, 10, 12, 14, 16, 0, 20, 22};   /* SURVEY 8c KAT-A */
    for (int r = 0; r < 20; ++r)
        for (int c = 0; c < 24; ++c)
            if (pinned[r * 24 + c] != (c == kat_cols[r])) { fprintf(stderr, "facade one-hot mismatch row %d col %d\n", r, c); return 10; }
    if (pinned[496] != 0) return 11;
    const uint8_t undo[4] = {0, 4, 1, 5};                      /* U R U' R' = (R U R' U')^-1: one launch */
    RC(rc_facade_steps(st, pitch, 3, undo, 4, pinned, 9u, 1, NULL));
    if (pinned[496] != 1) return 12;
    RC(rc_facade_step(st, pitch, 3, 12, pinned, 77u, 0, NULL));            /* wait = 0: the caller polls (here: a stream sync); 12 = no-op */
    CK(hipDeviceSynchronize());
    if (*(volatile uint32_t *)(pinned + 504) != 77u || pinned[496] != 1) return 15;
    RC(rc_facade_expand(st, pitch, 3, pinned, 10u, 1, 1, NULL));
    for (int a = 0; a < 12; ++a) if (pinned[288 + a] != 0) return 13;      /* no child of the solved cube is solved */
    /* per-call tuning override instead of any global knob */
    RC(rc_apply_moves_ex(st, st, act, n, pitch, pitch, 3, reward, done, NULL, RC_FMT_NONE, 0, NULL, 21));
    RC(rc_facade_release(pinned));
    CK(hipHostFree(pinned));
    /* hipHostRegister'ed memory: the kernel writes through the DEVICE alias of the buffer, the host polls the host address */
    void *raw = NULL;
    if (posix_memalign(&raw, 4096, 8192) != 0) return 18;
    memset(raw, 0, 8192);
    CK(hipHostRegister(raw, 8192, hipHostRegisterMapped));
    RC(rc_fill_solved(st, n, pitch, 3, NULL));
    RC(rc_facade_step(st, pitch, 3, 0, (uint8_t *)raw, 5u, 1, NULL));       /* U */
    if (((uint8_t *)raw)[496] != 0) return 19;
    RC(rc_facade_step(st, pitch, 3, 1, (uint8_t *)raw, 6u, 1, NULL));       /* U' -> solved */
    if (((uint8_t *)raw)[496] != 1) return 20;
    RC(rc_facade_release((const uint8_t *)raw));
    CK(hipHostUnregister(raw));
    free(raw);
    /* the workspace route from plain C: 2^17 solved cubes, all turned by U (action 0), dense float one-hot; every cube's row 0
     * must then carry its 1 where the one-launch kernel of rc_apply_moves puts it, and no cube is solved any more */
    {
        const int64_t m = (int64_t)1 << 17, mp = m;
        const int64_t need = rc_workspace_bytes(RC_OP_STEP, 3, m, RC_FMT_F32);
        if (need != 20 * m || rc_workspace_bytes(RC_OP_STEP, 3, 4096, RC_FMT_F32) != 0) return 21;
        uint8_t *big, *bact, *bdone, *ws;
        float *oh_a, *oh_b;
        CK(hipMalloc((void **)&big, 54 * mp)); CK(hipMalloc((void **)&bact, m)); CK(hipMalloc((void **)&bdone, m));
        CK(hipMalloc((void **)&ws, need)); CK(hipMalloc((void **)&oh_a, m * 480 * sizeof(float))); CK(hipMalloc((void **)&oh_b, m * 480 * sizeof(float)));
        CK(hipMemset(bact, 0, m));
        RC(rc_fill_solved(big, m, mp, 3, NULL));
        RC(rc_apply_moves_ws(big, big, bact, m, mp, mp, 3, NULL, bdone, oh_a, RC_FMT_F32, 0, ws, need, NULL));
        RC(rc_fill_solved(big, m, mp, 3, NULL));
        RC(rc_apply_moves(big, big, bact, m, mp, mp, 3, NULL, bdone, oh_b, RC_FMT_F32, 0, NULL));
        float *ha = (float *)malloc(480 * sizeof(float) * 2), *hb = ha + 480;
        for (int64_t c = 0; c < m; c += m / 7) {                 /* a few cubes across the batch, the last pass included below */
            CK(hipMemcpy(ha, oh_a + c * 480, 480 * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb, oh_b + c * 480, 480 * sizeof(float), hipMemcpyDeviceToHost));
            if (memcmp(ha, hb, 480 * sizeof(float)) != 0) return 22;
        }
        CK(hipMemcpy(ha, oh_a + (m - 1) * 480, 480 * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb, oh_b + (m - 1) * 480, 480 * sizeof(float), hipMemcpyDeviceToHost));
        float ones = 0;
        for (int i = 0; i < 480; ++i) ones += ha[i];
        if (memcmp(ha, hb, 480 * sizeof(float)) != 0 || ones != 20.0f) return 23;
        CK(hipMemcpy(hdone, bdone, 8, hipMemcpyDeviceToHost));
        if (hdone[0] != 0 || hdone[7] != 0) return 24;
        if (rc_apply_moves_ws(big, big, bact, m, mp, mp, 3, NULL, bdone, oh_a, RC_FMT_F32, 0, ws + 1, need, NULL) != RC_EINVAL) return 25;   /* misaligned */
        free(ha);
        CK(hipFree(big)); CK(hipFree(bact)); CK(hipFree(bdone)); CK(hipFree(ws)); CK(hipFree(oh_a)); CK(hipFree(oh_b));
    }
    /* round-5 entry points from plain C.  (1) reset(seed = 0, 5) draws 5, 0, 3, 11, 3 from both forms of the device generator */
    {
        uint32_t *seeds;
        uint8_t *draws, hd[5 * 256];
        const uint32_t zero = 0;
        CK(hipMalloc((void **)&seeds, 16)); CK(hipMalloc((void **)&draws, 5 * 256));
        CK(hipMemcpy(seeds, &zero, 4, hipMemcpyHostToDevice));
        const int forms[3] = {0, RC_VARIANT_LEGACY_LDS, RC_VARIANT_LEGACY_STREAM(0)};
        const uint8_t want[5] = {5, 0, 3, 11, 3};
        for (int f = 0; f < 3; ++f) {
            CK(hipMemset(draws, 0xEE, 5 * 256));
            RC(rc_legacy_scramble_actions_ex(seeds, NULL, 5, 5, 1, 3, draws, 256, NULL, forms[f]));
            CK(hipMemcpy(hd, draws, 5 * 256, hipMemcpyDeviceToHost));
            for (int d = 0; d < 5; ++d) if (hd[d * 256] != want[d]) { fprintf(stderr, "legacy draw %d form %d: %d\n", d, f, hd[d * 256]); return 26; }
        }
        if (rc_legacy_scramble_actions_ex(seeds, NULL, 5, 5, 1, 3, draws, 256, NULL, 3) != RC_EINVAL) return 27;
        if (rc_legacy_scramble_actions_ex(seeds, NULL, 5, 5, 1, 3, draws, 256, NULL, RC_VARIANT_LEGACY_STREAM(624)) != RC_EINVAL) return 27;
        CK(hipFree(seeds)); CK(hipFree(draws));
    }
    /* (2) rc_scramble_from replays R U R' U' that it READS from pinned host memory (rc_host_alias) out of untouched solved cubes;
     * (3) one expansion + rc_search_pack gives one record per cube; (4) a variant field of another entry point's group is refused */
    {
        uint8_t *paths, *alias = NULL, *work, *ccode, *csolved, *lcode, *leaf, *child, *solved, hrec[5 * 12 * 20];
        CK(hipHostMalloc((void **)&paths, 4 * 256, hipHostMallocMapped));
        for (int d = 0; d < 4; ++d) memset(paths + d * 256, seq[d], 256);
        RC(rc_host_alias(paths, (void **)&alias));
        if (rc_host_alias(hrec, (void **)&alias) != RC_EINVAL) return 28;                       /* plain stack memory is not host-mapped */
        RC(rc_host_alias(paths, (void **)&alias));
        CK(hipMalloc((void **)&work, 54 * pitch));
        RC(rc_fill_solved(st, n, pitch, 3, NULL));
        RC(rc_scramble_from(st, work, n, pitch, 3, 4, 0, 0, 0, alias, NULL, 256, done, NULL, NULL));
        CK(hipMemcpy(host, work, 54 * pitch, hipMemcpyDeviceToHost));
        for (int c = 0; c < n; ++c)
            for (int s2 = 0; s2 < 54; ++s2)
                if (host[s2 * pitch + c] != kat[s2] - '0') { fprintf(stderr, "scramble_from mismatch cube %d sticker %d\n", c, s2); return 29; }
        CK(hipMemcpy(host, st, 54 * pitch, hipMemcpyDeviceToHost));
        for (int s2 = 0; s2 < 54; ++s2) if (host[s2 * pitch] != s2 / 9) return 30;                /* the source states are untouched */
        CK(hipMalloc((void **)&ccode, 12 * 20 * pitch)); CK(hipMalloc((void **)&csolved, 12 * pitch)); CK(hipMalloc((void **)&lcode, 20 * pitch));
        CK(hipMalloc((void **)&leaf, n * 20)); CK(hipMalloc((void **)&child, n * 12 * 20)); CK(hipMalloc((void **)&solved, n * 12));
        RC(rc_expand_children(st, n, pitch, 3, NULL, csolved, ccode, pitch, NULL));
        RC(rc_encode(st, n, pitch, 3, lcode, RC_FMT_CODE, pitch, NULL));
        RC(rc_search_pack(lcode, ccode, csolved, n, pitch, 3, leaf, child, solved, NULL));
        CK(hipMemcpy(hrec, leaf, n * 20, hipMemcpyDeviceToHost));
        for (int c = 0; c < n; ++c)
            for (int p = 0; p < 20; ++p)                                                        /* solved cube: piece p in slot p, orientation 0 */
                if (hrec[c * 20 + p] != (p < 8 ? 3 * p : 2 * (p - 8))) { fprintf(stderr, "leaf code cube %d slot %d: %d\n", c, p, hrec[c * 20 + p]); return 31; }
        CK(hipMemcpy(hrec, solved, n * 12, hipMemcpyDeviceToHost));
        for (int i = 0; i < n * 12; ++i) if (hrec[i] != 0) return 32;
        CK(hipMemcpy(hrec, child, n * 12 * 20, hipMemcpyDeviceToHost));
        if (memcmp(hrec, hrec + 4 * 12 * 20, 12 * 20) != 0) return 33;                           /* every cube is the same cube */
        if (rc_apply_moves_ex(st, st, act, n, pitch, pitch, 3, reward, done, NULL, RC_FMT_NONE, 0, NULL, RC_VARIANT_ADI_SEGS(2)) != RC_EINVAL) return 34;
        if (rc_describe_dispatch(RC_OP_EXPAND, 3, 4096, 0, RC_OUT_FLAGS, RC_FMT_NONE, RC_VARIANT_STEP_POLICY(1), what, (int)sizeof what) != RC_EINVAL) return 35;
        RC(rc_describe_dispatch(RC_OP_FAMILY_TO_DENSE, 3, 200, 30, 0, RC_FMT_F32, 0, what, (int)sizeof what));
        if (!strstr(what, "family> depths=30")) { fprintf(stderr, "describe: %s\n", what); return 36; }
        CK(hipHostFree(paths)); CK(hipFree(work)); CK(hipFree(ccode)); CK(hipFree(csolved)); CK(hipFree(lcode)); CK(hipFree(leaf)); CK(hipFree(child)); CK(hipFree(solved));
    }
    uint32_t status = 99;
    RC(rc_read_status(&status, NULL));
    if (status != 0) return 8;
    if (rc_fill_solved(st, 1, (int64_t)1 << 27, 3, NULL) != RC_EINVAL) return 14;   /* rows * pitch must stay below 2^32 */
    if (rc_fill_solved(NULL, 1, 256, 3, NULL) != RC_EINVAL || strlen(rc_last_error()) == 0) return 9;
    puts("abi_smoke ok");
    return 0;
}
