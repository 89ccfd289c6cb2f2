"""Pointclouds -- the batch container the hot path's callers hold.

The reference's container (pytorch3d_pointops/structures/point_structure.py:40-1142)
is host-side bookkeeping that never calls the kernels; it is out of scope to
accelerate but its accessor contract IS the hot path's
input contract: ``points_padded()`` (:581), ``features_padded()`` (:605),
``num_points_per_cloud()`` (:623), ``cloud_to_packed_first_idx()`` (:645).  This is
a small from-scratch container with that contract: list / padded / packed views of
points (N clouds of P_n x 3) plus a ``dict[str, Tensor]`` of per-point features.

Unlike the reference, whose list->padded conversion loops over clouds in Python
(st/utils.py:19-79), the conversions here run on the device through the
packed<->padded HIP kernels when the data lives on a GPU.
"""
from typing import Dict, List, Optional, Sequence, Union

import torch

TensorOrList = Union[torch.Tensor, Sequence[torch.Tensor]]


def _to_padded(packed: torch.Tensor, first_idx: torch.Tensor, max_size: int) -> torch.Tensor:
    if packed.is_cuda and packed.dtype == torch.float32:
        from ..functions.packed_to_padded import packed_to_padded

        return packed_to_padded(packed, first_idx, int(max_size))
    # host tensors (the container itself is device-agnostic bookkeeping)
    N = first_idx.shape[0]
    out = packed.new_zeros((N, max_size) + tuple(packed.shape[1:]))
    ends = torch.cat([first_idx[1:], first_idx.new_tensor([packed.shape[0]])])
    for n in range(N):
        s, e = int(first_idx[n]), int(ends[n])
        out[n, : e - s] = packed[s:e]
    return out


def _to_packed(padded: torch.Tensor, first_idx: torch.Tensor, lengths: torch.Tensor, total: int) -> torch.Tensor:
    if padded.is_cuda and padded.dtype == torch.float32:
        from ..functions.packed_to_padded import padded_to_packed

        return padded_to_packed(padded, first_idx, int(total))
    return torch.cat([padded[n, : int(lengths[n])] for n in range(padded.shape[0])], dim=0) if padded.shape[0] else \
        padded.new_zeros((0,) + tuple(padded.shape[2:]))


class Pointclouds:
    """Batch of N point clouds with optional named per-point features.

    Args:
        points: list of (P_n, 3) tensors, or a padded (N, P, 3) tensor.
        features: None or ``{name: list of (P_n, C) tensors | padded (N, P, C) tensor}``.
    """

    def __init__(self, points: TensorOrList, features: Optional[Dict[str, TensorOrList]] = None) -> None:
        self._points_list: Optional[List[torch.Tensor]] = None
        self._points_padded: Optional[torch.Tensor] = None
        self._points_packed: Optional[torch.Tensor] = None
        self._feat_list: Dict[str, List[torch.Tensor]] = {}
        self._feat_padded: Dict[str, torch.Tensor] = {}
        self._feat_packed: Dict[str, torch.Tensor] = {}
        self.device = torch.device("cpu")

        if isinstance(points, (list, tuple)):
            self._points_list = list(points)
            self._N = len(self._points_list)
            if self._N > 0:
                self.device = self._points_list[0].device
            for p in self._points_list:
                if len(p) > 0 and (p.dim() != 2 or p.shape[1] != 3):
                    raise ValueError("Clouds in list must be of shape Px3 or empty")
                if p.device != self.device:
                    raise ValueError("All points must be on the same device")
            lens = [len(p) for p in self._points_list]
            self._num_points = torch.tensor(lens, dtype=torch.int64, device=self.device)
            self._P = max(lens) if lens else 0
        elif torch.is_tensor(points):
            if points.dim() != 3 or points.shape[2] != 3:
                raise ValueError("Points tensor has incorrect dimensions.")
            self._points_padded = points
            self._N, self._P = points.shape[0], points.shape[1]
            self.device = points.device
            self._num_points = torch.full((self._N,), self._P, dtype=torch.int64, device=self.device)
        else:
            raise ValueError("Points must be either a list or a tensor with shape (batch_size, P, 3).")

        self.valid = self._num_points > 0
        self.equisized = bool(self._N > 0 and int(self._num_points.min()) == self._P)
        cs = torch.cumsum(self._num_points, 0)
        self._first_idx = torch.cat([cs.new_zeros(1), cs[:-1]]) if self._N > 0 else cs
        self._total = int(cs[-1]) if self._N > 0 else 0

        for name, f in (features or {}).items():
            if f is None:
                continue
            if isinstance(f, (list, tuple)):
                f = list(f)
                if len(f) != self._N:
                    raise ValueError(f"feature '{name}' must have one entry per cloud")
                for n, t in enumerate(f):
                    if t.shape[0] != int(self._num_points[n]):
                        raise ValueError(f"feature '{name}' of cloud {n} has a wrong number of points")
                self._feat_list[name] = f
            elif torch.is_tensor(f):
                if f.dim() != 3 or f.shape[0] != self._N or f.shape[1] != self._P:
                    raise ValueError(f"feature '{name}' must be of shape (N, P, C)")
                self._feat_padded[name] = f
            else:
                raise ValueError(f"feature '{name}' must be a list or a padded tensor")
        self._feature_names = list((features or {}).keys())

    # ------------------------------------------------------------------ basic protocol
    def __len__(self) -> int:
        return self._N

    def isempty(self) -> bool:
        return self._N == 0 or not bool(self.valid.any())

    def feature_names(self) -> List[str]:
        return [n for n in self._feature_names if n in self._feat_list or n in self._feat_padded]

    def num_points_per_cloud(self) -> torch.Tensor:
        """(N,) int64 -- reference accessor st/point_structure.py:623."""
        return self._num_points

    def cloud_to_packed_first_idx(self) -> torch.Tensor:
        """(N,) int64 first packed row of each cloud -- reference accessor :645."""
        return self._first_idx

    def packed_to_cloud_idx(self) -> torch.Tensor:
        return torch.repeat_interleave(torch.arange(self._N, device=self.device), self._num_points)

    def padded_to_packed_idx(self) -> torch.Tensor:
        """Indices into the flattened (N*P) padded rows of every packed row -- reference :656."""
        within = torch.arange(self._total, device=self.device) - self._first_idx[self.packed_to_cloud_idx()]
        return self.packed_to_cloud_idx() * self._P + within

    # ------------------------------------------------------------------ points
    def points_list(self) -> List[torch.Tensor]:
        if self._points_list is None:
            pad = self.points_padded()
            self._points_list = [pad[n, : int(self._num_points[n])] for n in range(self._N)]
        return self._points_list

    def points_packed(self) -> torch.Tensor:
        if self._points_packed is None:
            if self._points_list is not None:
                self._points_packed = (torch.cat(self._points_list, dim=0) if self._N > 0
                                       else torch.zeros((0, 3), device=self.device))
            else:
                self._points_packed = _to_packed(self._points_padded, self._first_idx, self._num_points, self._total)
        return self._points_packed

    def points_padded(self) -> torch.Tensor:
        """(N, max P_n, 3) zero-padded -- reference accessor st/point_structure.py:581."""
        if self._points_padded is None:
            if self._N == 0:
                self._points_padded = torch.zeros((0, 0, 3), device=self.device)
            else:
                self._points_padded = _to_padded(self.points_packed(), self._first_idx, self._P)
        return self._points_padded

    # ------------------------------------------------------------------ features
    def features_list(self) -> Dict[str, List[torch.Tensor]]:
        out = {}
        for name in self.feature_names():
            if name not in self._feat_list:
                pad = self._feat_padded[name]
                self._feat_list[name] = [pad[n, : int(self._num_points[n])] for n in range(self._N)]
            out[name] = self._feat_list[name]
        return out

    def features_packed(self) -> Dict[str, torch.Tensor]:
        out = {}
        for name in self.feature_names():
            if name not in self._feat_packed:
                if name in self._feat_list:
                    self._feat_packed[name] = torch.cat(self._feat_list[name], dim=0)
                else:
                    self._feat_packed[name] = _to_packed(self._feat_padded[name], self._first_idx,
                                                         self._num_points, self._total)
            out[name] = self._feat_packed[name]
        return out

    def features_padded(self) -> Dict[str, torch.Tensor]:
        """{name: (N, max P_n, C)} (possibly empty dict) -- reference accessor :605."""
        out = {}
        for name in self.feature_names():
            if name not in self._feat_padded:
                self._feat_padded[name] = _to_padded(self.features_packed()[name], self._first_idx, self._P)
            out[name] = self._feat_padded[name]
        return out

    # ------------------------------------------------------------------ helpers
    def __getitem__(self, index) -> "Pointclouds":
        if isinstance(index, int):
            index = [index]
        elif isinstance(index, slice):
            index = list(range(self._N))[index]
        elif torch.is_tensor(index):
            index = index.nonzero().flatten().tolist() if index.dtype == torch.bool else index.tolist()
        pl = self.points_list()
        fl = self.features_list()
        return Pointclouds([pl[i] for i in index], {k: [v[i] for i in index] for k, v in fl.items()} or None)

    def to(self, device) -> "Pointclouds":
        device = torch.device(device)
        fl = self.features_list()
        return Pointclouds([p.to(device) for p in self.points_list()],
                           {k: [t.to(device) for t in v] for k, v in fl.items()} or None)

    def cuda(self) -> "Pointclouds":
        return self.to("cuda")

    def cpu(self) -> "Pointclouds":
        return self.to("cpu")

    def clone(self) -> "Pointclouds":
        fl = self.features_list()
        return Pointclouds([p.clone() for p in self.points_list()],
                           {k: [t.clone() for t in v] for k, v in fl.items()} or None)

    def detach(self) -> "Pointclouds":
        fl = self.features_list()
        return Pointclouds([p.detach() for p in self.points_list()],
                           {k: [t.detach() for t in v] for k, v in fl.items()} or None)


def join_pointclouds_as_batch(pointclouds: Sequence[Pointclouds]) -> Pointclouds:
    """Concatenate batches (reference: st/point_structure.py:1145)."""
    if not all(isinstance(pc, Pointclouds) for pc in pointclouds):
        raise ValueError("Wrong first argument to join_points_as_batch.")
    names = None
    for pc in pointclouds:
        n = set(pc.feature_names())
        names = n if names is None else names & n
    pts = [p for pc in pointclouds for p in pc.points_list()]
    feats = {k: [t for pc in pointclouds for t in pc.features_list()[k]] for k in sorted(names or [])}
    return Pointclouds(pts, feats or None)
