"""Pre-split weight operands for the x3 GEMM core (include/tris_hip.h: tris_weight_planes_f32, *_wp products).

In split-bf16 arithmetic every workgroup re-splits the tile of B it stages; for a weight matrix that work is identical in
every M tile of every launch.  `WeightPlanes` keeps, per weight, three bf16 planes (and the transposed -- for 3x3
convolutions also tap-mirrored -- planes that turn the data-gradient product into the same row-major form) and rebuilds
all of them with ONE table-driven launch.

Staleness is excluded by construction: planes are only consulted inside `WeightPlanes.active()`, and `train_step`
refreshes the trainable set at the top of every step (weights do not change between there and the end of backward);
frozen sets are built once and guarded by the parameters' version counters.
"""
import contextlib

import torch

from . import ops
from ._lib import call

_ACTIVE = 0   # nesting depth of WeightPlanes.active()


def lookup(w):
    """-> (P ptr, plane stride, PT ptr, PT plane stride) of a weight tensor, or None (no planes / not inside active())"""
    if _ACTIVE <= 0:
        return None
    wp = getattr(w, "_tris_wp", None)
    if wp is None or wp[4] != w._version:
        return None
    return wp


class WeightPlanes:
    def __init__(self, named_params, frozen=False):
        """named_params: iterable of (name, Parameter).  Eligible: 2-D / [N,K,1,1] matrices used by Linear / 1x1 conv and
        channels-last [Cout,Cin,3,3] conv weights, dimensions multiples of 8 (16-byte plane rows)."""
        self.frozen = frozen
        self.named = list(named_params)
        self._build()

    def _build(self):
        named_params = self.named
        self.items = []
        rows = []
        tiles = 0
        for name, p in named_params:
            if not p.is_cuda or p.dtype != torch.float32 or p.dim() not in (2, 4):
                continue
            if name.endswith("token_embedding.weight") or name.endswith("positional_embedding"):
                continue
            if p.dim() == 4 and tuple(p.shape[2:]) == (3, 3):
                if not p.is_contiguous(memory_format=torch.channels_last):
                    continue
                co, ci = p.shape[0], p.shape[1]
                if co % 8 or ci % 8:
                    continue
                n = p.numel()
                planes = torch.empty(3, n, dtype=torch.bfloat16, device=p.device)
                planes_t = torch.empty(3, n, dtype=torch.bfloat16, device=p.device)
                for tap in range(9):   # W[co][tap][ci] -> P same indexing; PT = Wd[ci][8 - tap][co]
                    rows.append([p.data_ptr() + 4 * tap * ci, 9 * ci, co, ci, planes.data_ptr() + 2 * tap * ci, n,
                                 planes_t.data_ptr() + 2 * (8 - tap) * co, 9 * co, n, tiles])
                    tiles += ((co + 31) // 32) * ((ci + 31) // 32)
            elif p.dim() == 2 or tuple(p.shape[2:]) == (1, 1):
                if p.dim() == 4 and not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                    continue
                if p.dim() == 2 and not p.is_contiguous():
                    continue
                nn_, kk = p.shape[0], p.shape[1]
                if nn_ % 8 or kk % 8:
                    continue
                n = p.numel()
                planes = torch.empty(3, n, dtype=torch.bfloat16, device=p.device)
                planes_t = torch.empty(3, n, dtype=torch.bfloat16, device=p.device)
                rows.append([p.data_ptr(), kk, nn_, kk, planes.data_ptr(), n, planes_t.data_ptr(), nn_, n, tiles])
                tiles += ((nn_ + 31) // 32) * ((kk + 31) // 32)
            else:
                continue
            self.items.append((p, planes, planes_t))
        self.total_tiles = tiles
        self.table = torch.tensor(rows, dtype=torch.int64, device="cuda") if rows else None
        self.entries = len(rows)
        self.ptrs = [p.data_ptr() for p, _, _ in self.items]
        self.built = False

    def ensure(self):
        """(re)build what is out of date: the table if a parameter's storage moved (e.g. a new optimiser arena adopted it),
        the planes of a frozen set if they were never built or a parameter was modified in place since"""
        if self.ptrs != [p.data_ptr() for p, _, _ in self.items]:
            self._build()
        if not self.built or any(getattr(p, "_tris_wp", (0,) * 5)[4] != p._version for p, _, _ in self.items):
            self.refresh()

    def refresh(self):
        if self.table is None:
            return
        if self.ptrs != [p.data_ptr() for p, _, _ in self.items]:
            self._build()
        self.built = True
        call("tris_weight_planes_f32", ops.P(self.table), self.entries, self.total_tiles, ops._stream())
        for p, planes, planes_t in self.items:
            p._tris_wp = (planes.data_ptr(), planes.shape[1], planes_t.data_ptr(), planes_t.shape[1], p._version,
                          planes, planes_t)

    @staticmethod
    @contextlib.contextmanager
    def active():
        global _ACTIVE
        _ACTIVE += 1
        try:
            yield
        finally:
            _ACTIVE -= 1

    def nbytes(self):
        return sum(a.numel() * 2 + b.numel() * 2 for _, a, b in self.items)
