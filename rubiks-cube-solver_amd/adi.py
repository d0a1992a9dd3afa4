"""Batched ADI (autodidactic iteration) sample generation: SURVEY.md section 8 rows 14-15 and N1.

Device work (librubikhip.so): the random walks and their 12-child expansion (rc_adi_generate_family: the
51-byte family record per state; 2x2x2: rc_adi_generate with codes), the dense one-hots the value net reads
(rc_onehot_from_family / rc_onehot_from_code) and the target assembly (rc_adi_targets).  The value net itself is the caller's unmodified torch module
(model.py:31-45), called once per depth on 13 * walks states instead of twice per sample.

Reference semantics kept (gym-cube/gym_cube/envs/cube_env.py:177-252):
  * sample (walk, d) = state after d moves, d = 1..depth, walks start from solved;
  * a solved child wins: target_value 1.0, target_policy = lowest solved action (:229-232);
  * otherwise target = max_a(value(child_a) - 1), first maximal action (:239-246);
  * error = |value(state) - target| * d**(-temperature), evaluated in double (:247-251).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from .tables import get_env_config


@torch.no_grad()
def adi_samples(model, cube_size, n_walks, depth, temperature, device="cuda", model_device=None, actions=None,
                seed=0, stream_id=0, walk_offset=0, dense_budget_bytes=1 << 30, want_state_dense=False, dense_dtype=None):
    """Generate n_walks x depth ADI samples.  Returns a dict of tensors on `device`, walk-major:

        state_code     uint8   [W, D, SLOTS]   compact one-hot code of the sample state
        state          uint8   [W, D, R, C]    dense one-hot (only if want_state_dense)
        target_value   float32 [W, D]
        target_policy  int32   [W, D]
        scramble_count int64   [W, D]          (= d, 1-based)
        error          float64 [W, D]
        actions        uint8   [W, D]          the move that led to the sample state

    actions: optional uint8 [W, D] moves to replay (e.g. the host's legacy numpy draws, which makes
    the samples those of the reference for the same global seed); None draws on the device.
    dense_dtype: dtype of the one-hot stream fed to `model` (default: float32, or the dtype of the model's first floating-point
    parameter when that is bfloat16 / float16); the returned values are float32 either way."""
    dev = torch.device(device)
    (R, C), A = get_env_config(cube_size)
    SL = ops.N_SLOTS[cube_size]
    mdev = torch.device(model_device) if model_device is not None else _module_device(model, dev)
    # the dense stream is written in the dtype the net computes in (bf16 / f16 halve the 13 * p * 480 elements per depth)
    ddtype = _module_dtype(model) if dense_dtype is None else dense_dtype
    per_walk = (A + 1) * R * C * torch.empty((), dtype=ddtype).element_size()
    chunk = max(1, min(n_walks, dense_budget_bytes // per_walk))
    chunk = min(n_walks, max(1024, chunk // 1024 * 1024)) if n_walks > 1024 else n_walks
    weights = [float(d) ** (-1 * temperature) for d in range(1, depth + 1)]  # cube_env.py:247, Python pow
    outs = {k: [] for k in ("state_code", "target_value", "target_policy", "error", "actions")}
    if want_state_dense:
        outs["state"] = []
    acts_all = None if actions is None else torch.as_tensor(actions, dtype=torch.uint8)
    for w0 in range(0, n_walks, chunk):
        wc = min(chunk, n_walks - w0)
        # power-of-two pitch: the A children (and the `depth` parents) of a chunk are then ONE tiled code buffer
        # of A * tiles (depth * tiles) tiles, so a single launch turns all of them into dense one-hots
        # 3x3x3: the generator emits the FAMILY record (51 shared look-ups per state instead of 13 x 20 picked codes) and ONE
        # rc_onehot_from_family launch per depth expands it to the 12 child blocks + the parent block; 2x2x2 keeps the codes
        fam = cube_size == 3
        pitch, bufs = ops.adi_buffers(wc, depth, cube_size, dev, pitch=1024 if wc <= 1024 else ops.ADI_TILE,
                                      parent_code=not fam, child_code=not fam, family=fam)
        p = bufs["actions_out"].shape[1]                      # padded walk count (tiles * pitch)
        tiles = p // pitch
        if fam:
            prow = torch.from_numpy(_lib.family_layout(cube_size)[1][A].astype(np.int64)).to(dev)
        a_in = None
        if acts_all is not None:
            a_host = torch.zeros((depth, p), dtype=torch.uint8)
            a_host[:, :wc] = acts_all[w0:w0 + wc].t()
            a_in = a_host.to(dev)
        ops.adi_generate(wc, depth, cube_size, pitch, dev, seed=seed, stream_id=stream_id, walk_offset=walk_offset + w0,
                         actions_in=a_in, **bufs)
        dense = torch.zeros(((A + 1) * p, R, C), dtype=ddtype, device=dev)             # A child blocks + the parents (pad columns stay 0)
        parent_code = bufs["family"].index_select(2, prow) if fam else bufs["parent_code"]   # [depth, tiles, SLOTS, pitch]
        tv = torch.empty((depth, wc), dtype=torch.float32, device=dev)
        tp = torch.empty((depth, wc), dtype=torch.int32, device=dev)
        err = torch.empty((depth, wc), dtype=torch.float64, device=dev)
        for d in range(depth):
            if fam:
                ops.onehot_from_family(bufs["family"][d], wc, cube_size, dense, block_stride=p)
            else:
                ops.onehot_from_code(bufs["child_code"][d].view(A * tiles, SL, pitch), A * p, cube_size, dense[:A * p])
                ops.onehot_from_code(bufs["parent_code"][d], wc, cube_size, dense[A * p:A * p + wc])
            v = model(dense[:A * p + wc].to(mdev))[0].reshape(-1).to(device=dev, dtype=torch.float32)
            child_value = v[:A * p].view(A, p)                   # exactly rc_adi_targets' [A][pitch] layout
            pv = v[A * p:].contiguous()
            w = torch.full((wc,), weights[d], dtype=torch.float64, device=dev)
            tv[d], tp[d], err[d] = ops.adi_targets(child_value.contiguous(), bufs["child_solved"][d], wc, cube_size, pv, w)
        outs["state_code"].append(torch.stack([ops.to_aos(parent_code[d], wc) for d in range(depth)], 1).contiguous())
        outs["actions"].append(bufs["actions_out"][:, :wc].t().contiguous())
        outs["target_value"].append(tv.t().contiguous())
        outs["target_policy"].append(tp.t().contiguous())
        outs["error"].append(err.t().contiguous())
        if want_state_dense:
            sd = torch.empty((depth * p, R, C), dtype=torch.uint8, device=dev)
            ops.onehot_from_code(parent_code.reshape(depth * tiles, SL, pitch), depth * p, cube_size, sd)
            outs["state"].append(sd.view(depth, p, R, C)[:, :wc].permute(1, 0, 2, 3).contiguous())
    _lib_status(dev)
    res = {k: torch.cat(v, 0) for k, v in outs.items()}
    res["scramble_count"] = torch.arange(1, depth + 1, dtype=torch.int64, device=dev).expand(n_walks, depth).contiguous()
    return res


def _lib_status(dev):
    if _lib.read_status(dev) & _lib.STATUS_BAD_ACTION:
        raise IndexError("action out of range")  # cube_env.py:86,96


def _module_dtype(model):
    try:
        for prm in model.parameters():
            if prm.dtype in (torch.bfloat16, torch.float16):
                return prm.dtype
            if prm.is_floating_point():
                return torch.float32
    except (AttributeError, TypeError):
        pass
    return torch.float32


def _module_device(model, default):
    try:
        return next(model.parameters()).device
    except (StopIteration, AttributeError, TypeError):
        return default


def samples_to_dicts(res, cube_size):
    """The reference's replay-buffer records (cube_env.py:193): one dict per sample, walk-major order.
    'state' is the dense one-hot as the reference types it: int64 [20,24] (py333.py:238) or
    float64 [7,21] (cube_env.py:143)."""
    state = res["state"].cpu().numpy()
    state = state.astype(np.int64) if cube_size == 3 else state.astype(np.float64)
    tv = res["target_value"].cpu().numpy()
    tp = res["target_policy"].cpu().numpy()
    err = res["error"].cpu().numpy()
    W, D = tv.shape
    for w in range(W):
        for d in range(D):
            yield {"state": state[w, d], "target_value": float(tv[w, d]), "target_policy": int(tp[w, d]),
                   "scramble_count": d + 1, "error": float(err[w, d])}
