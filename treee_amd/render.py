"""Mirror of the splat seam of the `render` crate: Camera3D (render/src/camera_3d.rs:8-18),
Camera3DGPU::new (:59-63), PointCloud / PointCloudProperty / PointCloudEnvironment
(render/src/point_cloud.rs:129-243) and the draw (`PointCloud::render`, :148-173), which here
returns the depth + id buffers instead of presenting a frame."""
import math

import numpy as np

from . import _lib
from .context import default_context, dev_f32, host_f32, is_device, ptr

FIELD_OF_VIEW = 45.0 * math.tau / 360.0  # treee/src/camera.rs:7
U32_MAX = 0xFFFFFFFF


class Camera3D:
    """render::Camera3D {aspect, fovy, near, far}; defaults from treee/src/camera.rs:20-25."""

    def __init__(self, aspect, fovy=FIELD_OF_VIEW, near=0.1, far=10_000.0):
        self.aspect, self.fovy, self.near, self.far = aspect, fovy, near, far

    def projection(self):
        out = np.zeros(16, np.float32)
        _lib.check(_lib.lib().treee_perspective(self.aspect, self.fovy, self.near, self.far, ptr(out)))
        return out  # column-major, like bytemuck::cast_slice of na::Matrix4


def view_from_transform(transform):
    """Camera3DGPU::new: view = transform.inverse().to_homogeneous() (camera_3d.rs:60).
    `transform` is a 4x4 row-major numpy matrix; returns column-major float32[16]."""
    inv = np.linalg.inv(np.asarray(transform, dtype=np.float64))
    return np.ascontiguousarray(inv.T.astype(np.float32)).reshape(16)


def translation(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def rotation_x(angle):
    c, s = math.cos(angle), math.sin(angle)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


class PointCloudEnvironment:
    """render::PointCloudEnvironment::new(state, min, max, scale); defaults treee/src/program.rs:156."""

    def __init__(self, min=0, max=U32_MAX, scale=0.1):
        self.min, self.max, self.scale = min, max, scale


class PointCloudProperty:
    """render::PointCloudProperty::new(state, &[u32])."""

    def __init__(self, data):
        if is_device(data):
            import torch

            self.data = data.contiguous() if data.dtype == torch.int32 else data.to(torch.int32).contiguous()
        else:
            self.data = np.ascontiguousarray(data, dtype=np.uint32)
        self.length = int(self.data.shape[0])


class PointCloud:
    """render::PointCloud::new(state, &[Point3<f32>]) + render()."""

    def __init__(self, vertices):
        self.buffer = dev_f32(vertices, 3) if is_device(vertices) else host_f32(vertices, 3)
        self.instances = int(self.buffer.shape[0])

    def render(self, property, view, proj, environment, width, height, ctx=None, octree=None, knn=None):
        """one draw of all instances; returns (depth[h,w] f32, id[h,w] u32).  `octree` (an Octree built from these very
        vertices): draw from its Morton-sorted records instead (treee_splat_octree_dev); `knn` (a KDTree built from them): from the
        records it holds in Hilbert order (treee_splat_knn_dev) -- the same frame, bit for bit, either way."""
        if property.length != 0:
            assert property.length == self.instances, f"{property.length} {self.instances}"  # point_cloud.rs:161-167
        view = host_f32(view).reshape(16)
        proj = host_f32(proj).reshape(16)
        env = environment
        if is_device(self.buffer):
            import torch

            c = ctx or default_context(self.buffer.device)
            depth = torch.empty((height, width), dtype=torch.float32, device=self.buffer.device)
            ids = torch.empty((height, width), dtype=torch.int32, device=self.buffer.device)
            if knn is not None:
                assert len(knn) == self.instances
                _lib.check(_lib.lib().treee_splat_knn_dev(c.handle, knn._h, ptr(property.data), ptr(view), ptr(proj), env.scale, env.min, env.max,
                                                          width, height, None, 0, 0, ptr(depth), ptr(ids)))
                return depth, ids
            if octree is not None:
                assert octree.n_points == self.instances
                _lib.check(_lib.lib().treee_splat_octree_dev(c.handle, octree._h, ptr(property.data), ptr(view), ptr(proj), env.scale, env.min, env.max,
                                                             width, height, None, 0, 0, ptr(depth), ptr(ids)))
                return depth, ids
            _lib.check(
                _lib.lib().treee_splat_dev(
                    c.handle, ptr(self.buffer), ptr(property.data), self.instances, ptr(view), ptr(proj), env.scale, env.min, env.max,
                    width, height, None, 0, 0, ptr(depth), ptr(ids),
                )
            )
            return depth, ids
        depth = np.zeros((height, width), np.float32)
        ids = np.zeros((height, width), np.uint32)
        _lib.check(
            _lib.lib().treee_splat(
                ptr(self.buffer), ptr(property.data), self.instances, ptr(view), ptr(proj), env.scale, env.min, env.max, width, height,
                ptr(depth), ptr(ids),
            )
        )
        return depth, ids


def eye_dome_dev(depth, strength, ctx=None):
    """render/src/eye_dome.wgsl over a device depth buffer; `strength` as set by the UI slider
    is mapped like render/src/eye_dome.rs:226-230."""
    import torch

    s = 1.0 - strength
    s = (0.1 if s < 0.1 else s) ** 6
    c = ctx or default_context(depth.device)
    h, w = depth.shape
    out = torch.empty_like(depth)
    _lib.check(_lib.lib().treee_eye_dome_dev(c.handle, ptr(depth), w, h, s, ptr(out)))
    return out
