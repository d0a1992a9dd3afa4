"""Device-resident replay sink for ADI samples: SURVEY.md section 8f row N1, second half.

The reference's `ReplayBuffer` (utils.py:203-270) is a deque of Python dicts, one per sample, each
holding a dense [20,24] int64 one-hot (3 840 bytes); `get_random_samples` appends 6 000 of them
per call (cube_env.py:193-194).  `TensorReplayBuffer` keeps the same surface --

    append(x) . get_prioritized_sample() . update(idx, error) . len() . buffer[idx] -> 5-tuple

-- over ring-buffer TENSORS: the state as its 20-byte (7-byte) compact code on the device, targets
and scramble counts beside it, errors in a host float64 array (np.random.choice needs them there).
Dense one-hots exist only for the prioritised sample of the current epoch and are produced by ONE
rc_onehot_from_code launch in get_prioritized_sample(); mini-batches are slices of that tensor.

Semantics kept from the reference (utils.py):
  * deque(maxlen=buf_size): index 0 is the OLDEST sample, appending beyond the capacity drops the oldest (:215-216, :253-260);
  * get_prioritized_sample (:245-251): every index when len <= sample_size, otherwise
    np.random.choice(arange(len), sample_size, replace=False, p=error / sum(error)) on numpy's global legacy generator --
    the same seed gives the same indices as the reference class;
  * __getitem__ (:226-243): (state, target_value, target_policy, scramble_count, memory_idx) with the dtypes
    torch.tensor() gives the reference's Python values (int64 / float64 state, float32 value, int64 policy, count and index);
  * update(idx, error) overwrites the error of LOGICAL index idx (:262-270).
`append` takes either one reference-style sample dict or a whole `adi.adi_samples` result (walk-major [W, D, ...] tensors,
appended in the reference's order: cube by cube, depth by depth)."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib, ops
from .tables import get_env_config


class TensorReplayBuffer(Dataset):
    def __init__(self, buf_size, sample_size, cube_size=3, device="cuda"):
        (self.R, self.C), self.A = get_env_config(cube_size)
        self.cube_size = cube_size
        self.SL = ops.N_SLOTS[cube_size]
        self.buf_size, self.sample_size = int(buf_size), int(sample_size)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RubikHipError("TensorReplayBuffer keeps its samples on a HIP device (no CPU fallback)")
        cap = self.buf_size
        self.code = torch.zeros((cap, self.SL), dtype=torch.uint8, device=self.device)
        self.target_value = torch.zeros(cap, dtype=torch.float32, device=self.device)
        self.target_policy = torch.zeros(cap, dtype=torch.int64, device=self.device)
        self.scramble_count = torch.zeros(cap, dtype=torch.int64, device=self.device)
        self.error_memory = np.zeros(cap, dtype=np.float64)          # host: the sampling probabilities are drawn there
        self._start, self._count = 0, 0
        self.prioritized_idx = None
        self._dense = None                                           # uint8 [len(prioritized_idx), R, C] of the current sample, a CACHE:
        #   dropped by every append (the ring may move under the logical indices), rebuilt on the next read, so a state always
        #   comes from the same live entry as its targets -- what the reference's deque gives (utils.py:226-243)

    # ------------------------------------------------------------------ deque bookkeeping
    def _phys(self, logical):
        return (self._start + logical) % self.buf_size

    @property
    def size(self):
        """Samples held (the reference's len(memory); len(self) is the prioritised sample's length, as there)."""
        return self._count

    def _push(self, code, tv, tp, sc, err):
        """Append M samples given as device tensors code [M, SL] u8, tv [M] f32, tp [M], sc [M], err numpy/tensor [M] f64."""
        m = code.shape[0]
        cap = self.buf_size
        err = np.asarray(err.detach().cpu().numpy() if isinstance(err, torch.Tensor) else err, dtype=np.float64).reshape(-1)
        if m > cap:                                                  # only the newest `cap` survive a deque(maxlen)
            code, tv, tp, sc, err = code[m - cap:], tv[m - cap:], tp[m - cap:], sc[m - cap:], err[m - cap:]
            m = cap
        first = (self._start + self._count) % cap                    # physical slot of the first new sample
        idx = (first + torch.arange(m, device=self.device)) % cap
        self.code[idx] = code.to(self.device, torch.uint8)
        self.target_value[idx] = tv.to(self.device, torch.float32)
        self.target_policy[idx] = tp.to(self.device, torch.int64)
        self.scramble_count[idx] = sc.to(self.device, torch.int64)
        self.error_memory[(first + np.arange(m)) % cap] = err
        self._dense = None                                           # logical indices may now name other entries
        over = self._count + m - cap
        if over > 0:
            self._start = (self._start + over) % cap
            self._count = cap
        else:
            self._count += m

    def append(self, x):
        """One reference-style sample dict (cube_env.py:193) or a whole adi.adi_samples result."""
        if "state_code" in x:
            return self.append_batch(x)
        state = np.asarray(x["state"])
        if state.shape != (self.R, self.C):
            raise ValueError(f"state must be a one-hot of shape {(self.R, self.C)}")
        code = torch.from_numpy(_code_of_dense(state, self.cube_size)).view(1, self.SL)
        self._push(code.to(self.device), torch.tensor([x["target_value"]], dtype=torch.float32), torch.tensor([int(x["target_policy"])]),
                   torch.tensor([int(x["scramble_count"])]), np.array([x["error"]], np.float64))

    def append_batch(self, res):
        """All samples of an adi_samples result, walk-major (the reference appends cube by cube, depth by depth)."""
        code = res["state_code"].reshape(-1, self.SL)
        self._push(code, res["target_value"].reshape(-1), res["target_policy"].reshape(-1), res["scramble_count"].reshape(-1),
                   res["error"].reshape(-1))

    # ------------------------------------------------------------------ the reference's surface
    def get_prioritized_sample(self):
        n = self._count
        if n <= self.sample_size:
            self.prioritized_idx = np.arange(n)
        else:
            err = self.error_memory[self._phys(np.arange(n))]
            # utils.py:249 divides by Python's sum(error): a strictly sequential float64 sum.  np.add.accumulate adds in the same order
            # (np.sum pairs the terms and rounds differently, which would move np.random.choice's picks): bit-equal, 37 ms -> 2 ms at 500k
            self.prob_memory = err / np.add.accumulate(err)[-1]
            self.prioritized_idx = np.random.choice(np.arange(n), self.sample_size, replace=False, p=self.prob_memory)
        self._dense = self._expand(self.prioritized_idx)
        return self.prioritized_idx

    def _expand(self, logical_idx):
        """Dense uint8 one-hots [B, R, C] of the given logical indices: one gather + one rc_onehot_from_code launch."""
        b = len(logical_idx)
        if b == 0:
            return torch.empty((0, self.R, self.C), dtype=torch.uint8, device=self.device)
        phys = torch.from_numpy(self._phys(np.asarray(logical_idx, dtype=np.int64))).to(self.device)
        tiles, pitch = ops._tile_shape(b, None)
        soa = torch.zeros((tiles * pitch, self.SL), dtype=torch.uint8, device=self.device)
        soa[:b] = self.code[phys]
        soa = soa.view(tiles, pitch, self.SL).permute(0, 2, 1).contiguous()          # [tiles, SLOTS, pitch]: RC_FMT_CODE layout
        dense = torch.empty((b, self.R, self.C), dtype=torch.uint8, device=self.device)
        ops.onehot_from_code(soa, b, self.cube_size, dense)
        return dense

    def _sample_dense(self):
        if self._dense is None:
            self._dense = self._expand(self.prioritized_idx)
        return self._dense

    def __len__(self):
        return len(self.prioritized_idx)

    def __getitem__(self, idx):
        memory_idx = int(self.prioritized_idx[idx])
        p = self._phys(memory_idx)
        state = self._sample_dense()[idx].to(torch.int64 if self.cube_size == 3 else torch.float64)   # py333.py:238 / cube_env.py:143
        return (state, self.target_value[p], self.target_policy[p], self.scramble_count[p],
                torch.tensor(memory_idx, device=self.device))

    def update(self, idx, error):
        """error_memory[idx] = error for a logical index (utils.py:262-270); idx / error may also be equal-length arrays."""
        if isinstance(idx, torch.Tensor):
            idx = idx.detach().cpu().numpy()
        if isinstance(error, torch.Tensor):
            error = error.detach().cpu().numpy()
        self.error_memory[self._phys(np.asarray(idx, dtype=np.int64))] = np.asarray(error, dtype=np.float64)

    # ------------------------------------------------------------------ the batched path
    def batches(self, batch_size, shuffle=True, dtype=torch.float32):
        """The mini-batches DataLoader(self, batch_size, shuffle) would yield (update_params, utils.py:296-303), without a
        Python call per sample: (state [B,R,C] `dtype`, target_value, target_policy, scramble_count, memory_idx).  The order comes
        from a DataLoader over the bare indices, so the same torch seed visits the samples in exactly the DataLoader's order."""
        from torch.utils.data import DataLoader

        mem = torch.from_numpy(np.asarray(self.prioritized_idx, dtype=np.int64))
        phys_all = torch.from_numpy(self._phys(np.asarray(self.prioritized_idx, dtype=np.int64))).to(self.device)
        # the DataLoader itself walks an index-only view of this buffer: same sampler, same draws from torch's generator
        order = DataLoader(_Indices(len(self)), batch_size=batch_size, shuffle=shuffle, collate_fn=lambda b: torch.as_tensor(b, dtype=torch.int64))
        dense = self._sample_dense()
        for i in order:
            d = i.to(self.device)
            p = phys_all[d]
            yield (dense[d].to(dtype), self.target_value[p], self.target_policy[p], self.scramble_count[p], mem[i].to(self.device))


class _Indices(Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return int(i)

    def __getitems__(self, idx):
        return list(idx)


def _code_of_dense(state, cube_size):
    """Compact code of ONE dense one-hot handed over by a reference-style caller (inverse of pos_to_state_3, py333.py:235-246,
    and of the 2x2x2 convention, cube_env.py:142-147): index bookkeeping on 480 numbers, no cube arithmetic."""
    s = np.asarray(state)
    if cube_size == 3:
        return np.argmax(s, axis=1).astype(np.uint8)                                  # row = slot, column = code
    code = np.zeros(7, np.uint8)
    for piece in range(7):                                                            # row = piece, column = slot*3 + ori
        col = int(np.argmax(s[piece]))
        code[col // 3] = piece * 3 + col % 3
    return code
