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

    def get_features_list(self, name: str) -> Optional[List[torch.Tensor]]:
        """One feature as a list of (P_n, C) tensors, None if the batch has no such feature (reference :408)."""
        return self.features_list().get(name)

    def get_features_packed(self, name: str) -> Optional[torch.Tensor]:
        """One feature packed to (sum P_n, C), or None (reference :516)."""
        return self.features_packed().get(name)

    def get_features_padded(self, name: str) -> Optional[torch.Tensor]:
        """One feature padded to (N, max P_n, C), or None (reference :591)."""
        return self.features_padded().get(name)

    # ------------------------------------------------------------------ single clouds, copies of clouds
    def get_cloud(self, index: int):
        """(points (P, 3), {name: (P, C)}) of one cloud (reference :938)."""
        if not isinstance(index, int):
            raise ValueError("Cloud index must be an integer.")
        if index < 0 or index > self._N:
            raise ValueError("Cloud index must be in the range [0, N) where N is the number of clouds in the batch.")
        return self.points_list()[index], {k: v[index] for k, v in self.features_list().items()}

    def extend(self, N: int) -> "Pointclouds":
        """Every cloud N times in a row (reference :883)."""
        if not isinstance(N, int):
            raise ValueError("N must be an integer.")
        if N <= 0:
            raise ValueError("N must be > 0.")
        rep = lambda ts: [t.clone() for t in ts for _ in range(N)]  # noqa: E731
        return Pointclouds(rep(self.points_list()), {k: rep(v) for k, v in self.features_list().items()} or None)

    def split(self, split_sizes: list) -> List["Pointclouds"]:
        """Consecutive sub-batches of the given sizes, like torch.split (reference :913)."""
        if not all(isinstance(x, int) for x in split_sizes):
            raise ValueError("Value of split_sizes must be a list of integers.")
        out, at = [], 0
        for n in split_sizes:
            out.append(self[at:at + n])
            at += n
        return out

    # ------------------------------------------------------------------ coordinates
    def _set_points(self, packed: torch.Tensor) -> "Pointclouds":
        """New coordinates for every point (packed order): the three views are rebuilt from them."""
        self._points_packed = packed
        self._points_list = list(packed.split(self._num_points.tolist(), 0))
        if self._points_padded is not None:
            self._points_padded = _to_padded(packed, self._first_idx, self._P) if self._N > 0 else self._points_padded
        return self

    def offset_(self, offsets_packed: torch.Tensor) -> "Pointclouds":
        """Translate in place by a (3,) vector or by one offset per packed point (reference :968)."""
        packed = self.points_packed()
        if offsets_packed.shape == (3,):
            offsets_packed = offsets_packed.expand_as(packed)
        if offsets_packed.shape != packed.shape:
            raise ValueError("Offsets must have dimension (all_p, 3).")
        return self._set_points(packed + offsets_packed)

    def scale_(self, scale) -> "Pointclouds":
        """Multiply the coordinates in place by a scalar or by one factor per cloud (reference :998)."""
        if not torch.is_tensor(scale):
            scale = torch.full((self._N,), scale, device=self.device)
        per_point = scale.to(self.device)[self.packed_to_cloud_idx()].unsqueeze(1)
        return self._set_points(per_point * self.points_packed())

    def update_padded(self, new_points_padded: torch.Tensor,
                      new_features_padded: Optional[Dict[str, torch.Tensor]] = None) -> "Pointclouds":
        """The same batch (cloud sizes, auxiliary index tensors) around new padded points and, optionally, new padded
        features; without new features the old ones are shared (reference :1025)."""
        def check(x, c):
            if x.shape[0] != self._N:
                raise ValueError("new values must have the same batch dimension.")
            if x.shape[1] != self._P:
                raise ValueError("new values must have the same number of points.")
            if c is not None and x.shape[2] != c:
                raise ValueError("new values must have the same number of channels.")

        check(new_points_padded, 3)
        if new_features_padded is not None:
            if not isinstance(new_features_padded, dict):
                raise ValueError("new_features_padded must be a dictionary")
            widths = {k: v.shape[2] for k, v in self.features_padded().items()}
            for k, v in new_features_padded.items():
                check(v, widths[k])
        new = Pointclouds(new_points_padded, new_features_padded)
        new._num_points, new._first_idx, new._total = self._num_points, self._first_idx, self._total
        new.valid, new.equisized = self.valid, self.equisized
        if new_features_padded is None:
            new._feat_list, new._feat_padded, new._feat_packed = self._feat_list, self._feat_padded, self._feat_packed
            new._feature_names = list(self._feature_names)
        return new

    def inside_box(self, box: torch.Tensor) -> torch.Tensor:
        """(sum P_n,) bool: packed points inside a (2, 3) box or inside their cloud's box of (N, 2, 3) (reference :1102)."""
        if box.dim() > 3 or box.dim() < 2:
            raise ValueError("Input box must be of shape (2, 3) or (N, 2, 3).")
        if box.dim() == 3 and box.shape[0] != 1 and box.shape[0] != self._N:
            raise ValueError("Input box dimension is incompatible with pointcloud size.")
        if box.dim() == 2:
            box = box[None]
        if (box[..., 0, :] > box[..., 1, :]).any():
            raise ValueError("Input box is invalid: min values larger than max values.")
        pts = self.points_packed()
        per_point = box.expand(pts.shape[0], 2, 3) if box.shape[0] == 1 else box[self.packed_to_cloud_idx()]
        return ((pts >= per_point[:, 0]) & (pts <= per_point[:, 1])).all(dim=-1)

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


def join_pointclouds_as_scene(pointclouds: Union[Pointclouds, List[Pointclouds]]) -> Pointclouds:
    """All clouds of a batch (or of a list of batches) as ONE cloud (reference: st/point_structure.py:1207)."""
    if isinstance(pointclouds, list):
        pointclouds = join_pointclouds_as_batch(pointclouds)
    if len(pointclouds) == 1:
        return pointclouds
    feats = {k: v[None] for k, v in pointclouds.features_packed().items()}
    return Pointclouds(pointclouds.points_packed()[None], feats or None)


def get_bounding_boxes(pointcloud: Pointclouds) -> torch.Tensor:
    """(N, 3, 2): min and max of every cloud along every axis (reference :1247)."""
    boxes = [torch.stack([p.min(dim=0)[0], p.max(dim=0)[0]], dim=1) for p in pointcloud.points_list()]
    return torch.stack(boxes, dim=0)


def offset(pointcloud: Pointclouds, offsets_packed: torch.Tensor) -> Pointclouds:
    """Out-of-place Pointclouds.offset_ (reference :1268)."""
    return pointcloud.clone().offset_(offsets_packed)


def scale(pointcloud: Pointclouds, scale: Union[float, torch.Tensor]) -> Pointclouds:  # noqa: A002
    """Out-of-place Pointclouds.scale_ (reference :1282)."""
    return pointcloud.clone().scale_(scale)


def subsample(pointclouds: Pointclouds, max_points: Union[int, Sequence[int]]) -> Pointclouds:
    """At most max_points (one value, or one per cloud) randomly kept points per cloud, features subsampled alike; the
    batch itself when nothing has to go.  Draws with numpy.random.choice like the reference (:1298)."""
    import numpy as np

    limits = [max_points] * len(pointclouds) if isinstance(max_points, int) else list(max_points)
    if len(limits) != len(pointclouds):
        raise ValueError("wrong number of max_points supplied")
    sizes = [int(n) for n in pointclouds.num_points_per_cloud()]
    if all(n <= int(m) for n, m in zip(sizes, limits)):
        return pointclouds
    pts, feats = [], {k: [] for k in pointclouds.feature_names()}
    fl = pointclouds.features_list()
    for i, (n, m, p) in enumerate(zip(sizes, limits, pointclouds.points_list())):
        keep = None
        if n > int(m):
            keep = torch.tensor(np.random.choice(n, int(m), replace=False), device=p.device, dtype=torch.int64)
        pts.append(p if keep is None else p[keep])
        for k in feats:
            feats[k].append(fl[k][i] if keep is None else fl[k][i][keep])
    return Pointclouds(pts, feats or None)


def all_close(pcd1: Pointclouds, pcd2: Pointclouds, rtol=1e-05, atol=1e-08, verbose=False) -> bool:
    """Same points and same features within tolerances (reference :1373)."""
    if pcd1.device != pcd2.device:
        raise ValueError("Pointclouds must be on the same device.")
    f1, f2 = pcd1.features_packed(), pcd2.features_packed()
    pts = torch.allclose(pcd1.points_packed(), pcd2.points_packed(), rtol, atol)
    if verbose:
        print("Points all close:", pts)
    if set(f1) != set(f2):
        if verbose:
            print("Features keys mismatch:", sorted(f1), sorted(f2))
        return False
    per = {k: torch.allclose(f1[k], f2[k], rtol, atol) for k in f1}
    if verbose:
        print("Features all close:", per)
    return pts and all(per.values())
