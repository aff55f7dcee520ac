"""Camera containers and the fly-around trajectory used by the sampling driver.

Stand-ins for the PyTorch3D pieces the reference driver uses
(/root/reference/holo_diffusion/utils/render_utils/flyaround.py:301-350,365-384):
``PerspectiveCameras`` (NDC convention, row vectors: ``X_cam = X_world R + T``),
``look_at_view_transform``, ``so3_exp_map`` and ``get_simple_360_camera_trajectory``.
Host-side fp32 math on tiny tensors; nothing here is on the device hot path.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch
import torch.nn.functional as F


class PerspectiveCameras:
    """Batch of pinhole cameras in PyTorch3D's NDC convention."""

    def __init__(self, R=None, T=None, focal_length=1.0, principal_point=((0.0, 0.0),), device="cpu"):
        R = torch.eye(3)[None] if R is None else torch.as_tensor(R, dtype=torch.float32)
        T = torch.zeros(1, 3) if T is None else torch.as_tensor(T, dtype=torch.float32)
        n = max(R.shape[0], T.shape[0])
        f = torch.as_tensor(focal_length, dtype=torch.float32)
        if f.dim() == 0:
            f = f.reshape(1, 1)
        if f.dim() == 1:
            f = f[:, None]
        pp = torch.as_tensor(principal_point, dtype=torch.float32).reshape(-1, 2)
        self.R = R.expand(n, 3, 3).clone().to(device)
        self.T = T.expand(n, 3).clone().to(device)
        self.focal_length = f.expand(n, f.shape[1]).clone().to(device)
        self.principal_point = pp.expand(n, 2).clone().to(device)

    def __len__(self):
        return self.R.shape[0]

    # A host-side copy of the (tiny) camera parameters travels with the object, so that handing device-resident
    # cameras to the renderer never costs a device->host synchronisation (the launch parameters are built on the host).
    def _host_key(self):
        # identity + in-place version of the four parameter tensors: attribute reassignment and in-place edits both
        # change it, so a stale host copy is never handed to the renderer
        return tuple((id(t), t._version) for t in (self.R, self.T, self.focal_length, self.principal_point))

    def host(self):
        """(R, T, focal_xy, principal_point) as CPU float32 tensors.  Cached against the identity and ``_version`` of
        the four tensors: reassigning an attribute or editing a tensor in place refreshes the copy on the next call
        (one device->host copy for device-resident cameras)."""
        h = self.__dict__.get("_host")
        key = self._host_key()
        if h is None or self.__dict__.get("_host_key_seen") != key:
            h = tuple(t.detach().to("cpu", torch.float32) for t in (self.R, self.T, self.focal_xy(), self.principal_point))
            self.__dict__["_host"] = h
            self.__dict__["_host_key_seen"] = key
        return h

    def invalidate_host(self) -> None:
        self.__dict__.pop("_host", None)

    def _adopt_host(self, host) -> None:
        self.__dict__["_host"] = host
        self.__dict__["_host_key_seen"] = self._host_key()

    def __getitem__(self, idx):
        if isinstance(idx, int):
            idx = [idx]
        had = self.__dict__.get("_host")
        c = PerspectiveCameras.__new__(PerspectiveCameras)
        c.R, c.T = self.R[idx], self.T[idx]
        c.focal_length, c.principal_point = self.focal_length[idx], self.principal_point[idx]
        if had is not None or not self.R.is_cuda:
            c._adopt_host(tuple(t[idx] for t in self.host()))
        return c

    def to(self, device):
        host = self.host()  # taken BEFORE the move: free for CPU cameras, one sync for device cameras
        c = PerspectiveCameras.__new__(PerspectiveCameras)
        c.R, c.T = self.R.to(device), self.T.to(device)
        c.focal_length, c.principal_point = self.focal_length.to(device), self.principal_point.to(device)
        c._adopt_host(host)
        return c

    @property
    def device(self):
        return self.R.device

    def get_camera_center(self) -> torch.Tensor:
        return -torch.bmm(self.T[:, None, :], self.R.transpose(1, 2))[:, 0]

    def focal_xy(self) -> torch.Tensor:
        f = self.focal_length
        return f.expand(-1, 2) if f.shape[1] == 1 else f


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees: bool = True,
                           up: Sequence[Sequence[float]] = ((0.0, 1.0, 0.0),)) -> Tuple[torch.Tensor, torch.Tensor]:
    dist = torch.as_tensor(dist, dtype=torch.float32).reshape(-1)
    elev = torch.as_tensor(elev, dtype=torch.float32).reshape(-1)
    azim = torch.as_tensor(azim, dtype=torch.float32).reshape(-1)
    if degrees:
        elev = math.pi / 180.0 * elev
        azim = math.pi / 180.0 * azim
    C = torch.stack([dist * torch.cos(elev) * torch.sin(azim), dist * torch.sin(elev),
                     dist * torch.cos(elev) * torch.cos(azim)], dim=1)
    at = torch.zeros_like(C)
    upv = torch.tensor(up, dtype=torch.float32).expand_as(C)
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(upv, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        x_axis = torch.where(is_close, F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5), x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


def so3_exp_map(log_rot: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    x, y, z = log_rot.unbind(1)
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).reshape(-1, 3, 3)
    return fac1[:, None, None] * K + fac2[:, None, None] * torch.bmm(K, K) + torch.eye(3)[None]


def get_simple_360_camera_trajectory(max_angle: float, n_flyaround_poses: int, camera_elevation: float,
                                     hemispherical_radius: float, up: Tuple[float, float, float],
                                     camera_focal_length: float,
                                     canonical_up: Tuple[float, float, float] = (0.0, -1.0, 0.0)) -> PerspectiveCameras:
    """flyaround.py:301-350 (angles in radians, like the reference signature)."""
    max_angle_deg = 360 * max_angle / (math.pi * 2)
    elev_deg = 360 * camera_elevation / (math.pi * 2)
    azimuths = torch.linspace(0, max_angle_deg, n_flyaround_poses + 1)[:n_flyaround_poses]
    rots, trans = [], []
    for az in azimuths:
        R, T = look_at_view_transform(dist=hemispherical_radius, elev=elev_deg, azim=float(az), up=(canonical_up,))
        rots.append(R)
        trans.append(T)
    rots, trans = torch.cat(rots, dim=0), torch.cat(trans, dim=0)
    axis = torch.cross(torch.tensor(canonical_up, dtype=torch.float32), torch.tensor(up, dtype=torch.float32), dim=0)
    R_plane = so3_exp_map(axis[None])[0]
    rots = torch.bmm(R_plane[None].expand_as(rots), rots)
    return PerspectiveCameras(R=rots, T=trans, focal_length=torch.ones(n_flyaround_poses, 1) * camera_focal_length,
                              principal_point=torch.zeros(n_flyaround_poses, 2))
