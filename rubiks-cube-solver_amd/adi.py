"""Batched ADI (autodidactic iteration) sample generation: SURVEY.md section 8 rows 14-15 and N1.

Device work (librubikhip.so): the random walks and their 12-child expansion (rc_adi_generate_family: the 51-byte family record
per state; 2x2x2: rc_adi_generate with codes), the dense one-hots the value net reads (rc_onehot_from_family_depths /
rc_onehot_from_code_blocks) and the target assembly (rc_adi_targets_depths).  The value net itself is the caller's unmodified torch module
(model.py:31-45).

Shape of one call (AdiPlan): ONE generator launch per chunk of walks, then per GROUP of depths one launch that writes the
[depths][A + 1][walks] one-hot input of the net, one forward of the net on that whole block and one launch that assembles the targets
of every (walk, depth) of the group straight from the net's output into walk-major result tensors.  The blocks are packed (the block
stride is the walk count rounded up to 8, not a padded tile count), and a group holds as many depths as the dense budget allows: the
reference's own size, 200 walks x depth 30 (config/config.yaml:7-8, called every epoch by train.py:152-155), is one group -- one
forward on 78 000 states instead of 30 forwards on 12 488 mostly padded rows.  `graph=True` captures everything behind the generator
launch as a hipGraph over the plan's static buffers and replays it on later calls.

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


class AdiPlan:
    """Buffers and launch sequence of adi_samples for ONE shape (model, cube size, walks, depth, temperature, dtype).

    run() returns a dict of tensors on `device`, walk-major:

        state_code     uint8   [W, D, SLOTS]   compact one-hot code of the sample state
        state          uint8   [W, D, R, C]    dense one-hot (only if want_state_dense)
        target_value   float32 [W, D]
        target_policy  int32   [W, D]
        scramble_count int64   [W, D]          (= d, 1-based)
        error          float64 [W, D]
        actions        uint8   [W, D]          the move that led to the sample state

    These are the plan's OWN tensors, overwritten by the next run(): pass clone=True (adi_samples does) to keep them."""

    def __init__(self, model, cube_size, n_walks, depth, temperature, device="cuda", model_device=None, dense_budget_bytes=1 << 30,
                 want_state_dense=False, dense_dtype=None, graph=False):
        self.model, self.cube_size, self.W, self.D = model, cube_size, int(n_walks), int(depth)
        self.dev = dev = torch.device(device)
        if dev.index is None and dev.type == "cuda":
            self.dev = dev = torch.device("cuda", torch.cuda.current_device())
        (self.R, self.C), self.A = get_env_config(cube_size)
        self.SL = ops.N_SLOTS[cube_size]
        self.mdev = torch.device(model_device) if model_device is not None else _module_device(model, dev)
        if self.mdev.type == "cuda" and self.mdev.index is None:
            self.mdev = torch.device("cuda", torch.cuda.current_device())
        # the dense stream is written in the dtype the net computes in (bf16 / f16 halve the 13 * walks * 480 elements per depth)
        self.ddtype = _module_dtype(model) if dense_dtype is None else dense_dtype
        self.want_state_dense = bool(want_state_dense)
        self.fam = cube_size == 3        # 3x3x3: the FAMILY record + one block-writing launch; 2x2x2 keeps parent / child codes
        W, D, A = self.W, self.D, self.A
        row_bytes = self.R * self.C * torch.empty((), dtype=self.ddtype).element_size()               # one dense one-hot
        chunk = max(1, min(W, dense_budget_bytes // ((A + 1) * row_bytes)))                           # walks whose single depth fits the budget
        self.chunk = min(W, max(1024, chunk // 1024 * 1024)) if W > 1024 else W
        rows_of = lambda wc: self._geometry(wc)[2]
        self.group = max(1, min(D, dense_budget_bytes // max(1, rows_of(self.chunk) * row_bytes)))   # depths per forward of the net
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        self.out = {"state_code": e((W, D, self.SL), torch.uint8), "target_value": e((W, D), torch.float32),
                    "target_policy": e((W, D), torch.int32), "error": e((W, D), torch.float64), "actions": e((W, D), torch.uint8),
                    "scramble_count": torch.arange(1, D + 1, dtype=torch.int64, device=dev).expand(W, D).contiguous()}
        if self.want_state_dense:
            self.out["state"] = e((W, D, self.R, self.C), torch.uint8)
        self.weights = torch.tensor([float(d) ** (-1 * temperature) for d in range(1, D + 1)], dtype=torch.float64, device=dev)  # cube_env.py:247, Python pow
        if self.fam:
            self.prow = torch.from_numpy(_lib.family_layout(cube_size)[1][A].astype(np.int64)).to(dev)
        self.chunks = []                                           # (first walk, walks, device buffers, pinned staging of replayed moves)
        by_size = {}                                               # chunks of one size share their device buffers (stream-ordered)
        for w0 in range(0, W, self.chunk if W else 1):
            wc = min(self.chunk, W - w0)
            if wc not in by_size:
                by_size[wc] = self._chunk_buffers(wc)
            self.chunks.append((w0, wc, by_size[wc], [None]))
        # zeros once: the pad rows between a block's walks and its stride are never written and must stay finite for the net
        self.dense = torch.zeros((self.group * rows_of(self.chunk) if W and D else 0, self.R, self.C), dtype=self.ddtype, device=dev)
        self.graph = bool(graph)
        self._graphs, self._graph_sig = {}, None
        if self.graph and self.mdev != dev:
            raise ValueError("AdiPlan(graph=True) needs the model on the cubes' device (a host model cannot be captured)")

    # ------------------------------------------------------------------ geometry
    def _geometry(self, wc):
        """(pitch of the generator's buffers, block stride, dense rows per depth) for a chunk of wc walks."""
        if self.fam:                                               # blocks are addressed by a stride: pack them
            pitch = _lib.pitch_for(wc) if wc <= ops.ADI_TILE else ops.ADI_TILE
            bs = -(-wc // 8) * 8
            return pitch, bs, (self.A + 1) * bs
        # 2x2x2 (no family record): the A child-code buffers of every depth and the parent-code buffers are equally tiled, so ONE
        # rc_onehot_from_code_blocks launch each packs them into blocks of ceil16(wc) rows ([depth][A][bs] children, then [depth][bs]
        # parents) -- round 5 fed the net blocks padded to whole tiles of >= 512 walks (200 walks: 2.56 x the rows)
        pitch = _lib.pitch_for(wc) if wc <= ops.ADI_TILE else ops.ADI_TILE
        bs = -(-wc // 16) * 16
        return pitch, bs, (self.A + 1) * bs

    def _chunk_buffers(self, wc):
        pitch, bs, _ = self._geometry(wc)
        pitch, bufs = ops.adi_buffers(wc, self.D, self.cube_size, self.dev, pitch=pitch, parent_code=not self.fam, child_code=not self.fam,
                                      family=self.fam)
        p = bufs["actions_out"].shape[1]                           # padded walk count (tiles * pitch)
        b = {"pitch": pitch, "bs": bs, "p": p, "tiles": p // pitch, "bufs": bufs}
        b["actions_in"] = torch.zeros((self.D, p), dtype=torch.uint8, device=self.dev)
        return b

    # ------------------------------------------------------------------ one chunk, behind the generator launch
    def _after_generate(self, w0, wc, b):
        A, D, SL, cs, dev = self.A, self.D, self.SL, self.cube_size, self.dev
        bufs, bs, p, tiles, pitch = b["bufs"], b["bs"], b["p"], b["tiles"], b["pitch"]
        tv, tp, err = (self.out[k][w0:] for k in ("target_value", "target_policy", "error"))
        for g0 in range(0, D, self.group):
            gc = min(self.group, D - g0)
            rows = gc * (A + 1) * bs
            x = self.dense[:rows]
            if self.fam:
                ops.onehot_from_family(bufs["family"][g0:g0 + gc], wc, cs, x, block_stride=bs, n_depths=gc)
            else:
                ops.onehot_from_code_blocks(bufs["child_code"][g0:g0 + gc].view(gc * A, tiles, SL, pitch), wc, cs, x[:gc * A * bs], bs)
                ops.onehot_from_code_blocks(bufs["parent_code"][g0:g0 + gc], wc, cs, x[gc * A * bs:], bs)
            v = self.model(x if self.mdev == dev else x.to(self.mdev))[0].reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
            if self.fam:                                           # [depth][A children, parent][bs]
                cv, cvd, pv, pvd = v, (A + 1) * bs, v[A * bs:], (A + 1) * bs
            else:                                                  # [depth][A][bs] children, then [depth][bs] parents
                cv, cvd, pv, pvd = v, A * bs, v[gc * A * bs:], bs
            ops.adi_targets_depths(cv, cvd, bs, bufs["child_solved"][g0:g0 + gc], pv, pvd, self.weights[g0:g0 + gc], wc, gc, cs,
                                   tv[:, g0:], tp[:, g0:], err[:, g0:])
        # walk-major copies of the generator's rows: parent codes (the sample states) and the moves
        pc = bufs["family"].index_select(2, self.prow) if self.fam else bufs["parent_code"]          # [D, tiles, SL, pitch]
        src = pc.permute(1, 3, 0, 2)                                                                 # [tiles, pitch, D, SL]
        sc = self.out["state_code"]
        full = wc // pitch
        if full:
            sc[w0:w0 + full * pitch].view(full, pitch, D, SL).copy_(src[:full])
        if wc > full * pitch:
            sc[w0 + full * pitch:w0 + wc].copy_(src[full, :wc - full * pitch])
        self.out["actions"][w0:w0 + wc].copy_(bufs["actions_out"][:, :wc].t())
        if self.want_state_dense:                                  # the reference's per-sample dicts only: one launch per depth
            sd = torch.empty((D, -(-wc // 16) * 16, self.R, self.C), dtype=torch.uint8, device=dev)   # every depth's block 16-byte aligned
            for d in range(D):
                ops.onehot_from_code(pc[d], wc, cs, sd[d])
            self.out["state"][w0:w0 + wc].copy_(sd[:, :wc].permute(1, 0, 2, 3))

    @torch.no_grad()
    def run(self, actions=None, seed=0, stream_id=0, walk_offset=0, clone=False):
        """actions: optional uint8 [W, D] moves to replay (e.g. the host's legacy numpy draws, which makes the samples those of the
        reference for the same global seed); None draws on the device from (seed, stream_id, walk_offset + walk)."""
        dev = self.dev
        acts_all = None
        if actions is not None:
            acts_all = torch.as_tensor(actions, dtype=torch.uint8)
            if tuple(acts_all.shape) != (self.W, self.D):
                raise ValueError(f"actions must be [{self.W}, {self.D}]")
        for w0, wc, b, stage in self.chunks if self.D else ():
            a_in = None
            if acts_all is not None:
                if acts_all.is_cuda:
                    b["actions_in"][:, :wc].copy_(acts_all[w0:w0 + wc].t())
                else:                                              # the chunk's own pinned staging buffer (the last run ended with a
                    if stage[0] is None:                           # synchronisation, so it is free), one asynchronous upload
                        stage[0] = torch.zeros((self.D, b["p"]), dtype=torch.uint8).pin_memory()
                    stage[0][:, :wc].copy_(acts_all[w0:w0 + wc].t())
                    b["actions_in"].copy_(stage[0], non_blocking=True)
                a_in = b["actions_in"]
            ops.adi_generate(wc, self.D, self.cube_size, b["pitch"], dev, seed=seed, stream_id=stream_id, walk_offset=walk_offset + w0,
                             actions_in=a_in, **b["bufs"])
            if not self.graph:
                self._after_generate(w0, wc, b)
                continue
            sig = _param_addresses(self.model)                 # a captured graph holds the parameters' ADDRESSES: in-place updates
            if sig != self._graph_sig:                         # (optimizer steps, load_state_dict) are seen, a re-allocation
                self._graphs.clear()                           # (model.to(...), .half()) is not -- capture again
                self._graph_sig = sig
            g = self._graphs.get(w0)
            if g is None:                                          # warm-up on a side stream, then capture (as rollout.py does)
                s = torch.cuda.Stream(dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    self._after_generate(w0, wc, b)
                torch.cuda.current_stream(dev).wait_stream(s)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._after_generate(w0, wc, b)
                self._graphs[w0] = g
            g.replay()
        _lib_status(dev)
        return {k: (v.clone() if clone else v) for k, v in self.out.items()}


_plans = {}          # adi_samples(graph=True): captured plans per call shape, so that repeated calls replay instead of re-capturing
MAX_KEPT_PLANS = 4   # each holds a reference to its model, its static buffers (dense: up to dense_budget_bytes) and the graph's private pool


def adi_samples(model, cube_size, n_walks, depth, temperature, device="cuda", model_device=None, actions=None,
                seed=0, stream_id=0, walk_offset=0, dense_budget_bytes=1 << 30, want_state_dense=False, dense_dtype=None, graph=False):
    """Generate n_walks x depth ADI samples (see AdiPlan for the result dict, the semantics and the launch sequence).

    actions: optional uint8 [W, D] moves to replay; None draws on the device.
    dense_dtype: dtype of the one-hot stream fed to `model` (default: float32, or the dtype of the model's first floating-point
    parameter when that is bfloat16 / float16); the returned values are float32 either way.
    graph: keep the plan of this call shape (static buffers + the captured hipGraph of everything behind the generator launch) in a
    module-level cache and replay it on the next call with the same model object and shape; results are copies either way.  The cache
    holds the MAX_KEPT_PLANS most recently used plans (each pins its model, up to dense_budget_bytes of device memory and the captured
    graph's activations); release_plans() drops them all."""
    if not graph:
        return AdiPlan(model, cube_size, n_walks, depth, temperature, device, model_device, dense_budget_bytes, want_state_dense,
                       dense_dtype).run(actions, seed, stream_id, walk_offset)
    key = (id(model), cube_size, int(n_walks), int(depth), float(temperature), str(torch.device(device)), str(model_device), int(dense_budget_bytes),
           bool(want_state_dense), str(dense_dtype), str(_module_dtype(model)), str(_module_device(model, None)))   # a plan freezes the net's dtype and device
    plan = _plans.pop(key, None)                      # re-inserted below: the dict's order is the order of last use (LRU)
    if plan is not None and plan.model is not model:
        plan = None
    if plan is None:
        while len(_plans) >= MAX_KEPT_PLANS:
            _plans.pop(next(iter(_plans)))            # the least recently used plan, not the oldest one: the hot shape stays captured
        plan = AdiPlan(model, cube_size, n_walks, depth, temperature, device, model_device, dense_budget_bytes, want_state_dense,
                       dense_dtype, graph=True)
    _plans[key] = plan
    return plan.run(actions, seed, stream_id, walk_offset, clone=True)


def release_plans():
    """Drop the plans adi_samples(graph=True) keeps (their device buffers and captured graphs)."""
    _plans.clear()


def _lib_status(dev):
    if _lib.read_status(dev) & _lib.STATUS_BAD_ACTION:
        raise IndexError("action out of range")  # cube_env.py:86,96


def _param_addresses(model):
    try:
        return tuple((prm.data_ptr(), prm.dtype) for prm in model.parameters())
    except (AttributeError, TypeError):
        return ()


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
