"""Camera-pose lists: mirrors of src/training/views_dataset.py (circle_poses :75-85, Zero123PlusDataset :88-149,
MultiviewDataset :151-218).  The list of views is what gets sharded one-per-GPU (dist.py)."""
import math
import numpy as np
import torch
from .utils import get_view_direction


def circle_poses(device, radius=1.25, theta=60.0, phi=0.0, angle_overhead=30.0, angle_front=60.0):
    theta = np.deg2rad(theta)
    phi = np.deg2rad(phi)
    angle_overhead = np.deg2rad(angle_overhead)
    angle_front = np.deg2rad(angle_front)
    thetas = torch.FloatTensor([theta])
    phis = torch.FloatTensor([phi])
    dirs = get_view_direction(thetas, phis, angle_overhead, angle_front)
    return dirs, thetas.item(), phis.item(), radius


class _PoseList:
    def collate(self, index):
        phi, theta = self.phis[index[0]], self.thetas[index[0]]
        dirs, thetas, phis, radius = circle_poses(self.device, radius=self.cfg.radius, theta=theta, phi=phi,
                                                  angle_overhead=self.cfg.overhead_range, angle_front=self.cfg.front_range)
        return {'dir': dirs, 'theta': thetas, 'phi': phis, 'radius': radius, 'base_theta': math.radians(self.cfg.base_theta)}

    def __len__(self):
        return self.size

    def __iter__(self):
        for i in range(self.size):
            yield self.collate([i])

    def dataloader(self):
        loader = list(self)
        return _Loader(loader, self)


class _Loader(list):
    def __init__(self, items, data):
        super().__init__(items)
        self._data = data


class Zero123PlusDataset(_PoseList):
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self.phis = [0] + [30, 150, 270, 90, 210, 330]
        self.thetas = [90 - t for t in ([30] + [30, 30, 30, -20, -20, -20])]
        self.size = len(self.phis)


class MultiviewDataset(_PoseList):
    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        size = cfg.n_views
        self.phis = [(index / size) * 360 for index in range(size)]
        self.thetas = [cfg.base_theta for _ in range(size)]
        alt = lambda l: [l[0]] + [i for j in zip(l[1:size // 2], l[-1:size // 2:-1]) for i in j] + [l[size // 2]]
        if cfg.alternate_views:
            self.phis, self.thetas = alt(self.phis), alt(self.thetas)
        for phi, theta in cfg.views_before:
            self.phis, self.thetas = [phi] + self.phis, [theta] + self.thetas
        for phi, theta in cfg.views_after:
            self.phis, self.thetas = self.phis + [phi], self.thetas + [theta]
        self.size = len(self.phis)


class ViewsDataset(_PoseList):
    """Evaluation orbit (src/training/views_dataset.py:220-260, the non-random branch): `size` views on a circle at
    radius 1.2 x cfg.radius, theta = cfg.base_theta, phi = index / size * 360 degrees."""
    def __init__(self, cfg, device, size=100, random_views=False):
        if random_views:
            raise NotImplementedError("ViewsDataset(random_views=True): rand_poses is unused on the paint / eval path (SURVEY section 2)")
        self.cfg, self.device, self.size, self.random_views = cfg, device, size, False

    def collate(self, index):
        phi = (index[0] / self.size) * 360
        dirs, thetas, phis, radius = circle_poses(self.device, radius=self.cfg.radius * 1.2, theta=self.cfg.base_theta, phi=phi,
                                                  angle_overhead=self.cfg.overhead_range, angle_front=self.cfg.front_range)
        return {'dir': dirs, 'theta': thetas, 'phi': phis, 'radius': radius, 'base_theta': math.radians(self.cfg.base_theta)}
