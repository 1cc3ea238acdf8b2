"""Host helpers of the painting loop: mirrors of the functions the hot path calls in src/utils.py
(get_view_direction :13-40, seed_everything :73-78, get_nonzero_region_tuple :92-113,
split/merge 3x2 grid :326-371) and the latent/image scale helpers of src/training/trainer.py:38-52.
Pure host/torch logic — pinned by tests/golden (outputs of the reference's own functions)."""
import os
import random
import numpy as np
import torch


def get_view_direction(thetas, phis, overhead, front):
    # front 0 | side(left) 1 | back 2 | side(right) 3 | top 4 | bottom 5   (utils.py:13-40; the "front" test can
    # never be true — res is zero-initialised, so front views still come out as 0: SURVEY Appendix B)
    res = torch.zeros(thetas.shape[0], dtype=torch.long)
    res[(phis >= (2 * np.pi - front / 2)) & (phis < front / 2)] = 0
    res[(phis >= front / 2) & (phis < (np.pi - front / 2))] = 1
    res[(phis >= (np.pi - front / 2)) & (phis < (np.pi + front / 2))] = 2
    res[(phis >= (np.pi + front / 2)) & (phis < (2 * np.pi - front / 2))] = 3
    res[thetas <= overhead] = 4
    res[thetas >= (np.pi - overhead)] = 5
    return res


def seed_everything(seed):
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def get_nonzero_region_tuple(mask):
    """Square-ified 1.1x bounding box of the nonzero region, clamped -> (min_h, min_w, max_h, max_w) python ints.
    The bbox is reduced on the device (row/col any + arg extremes) and read back once (4 ints) instead of
    materialising nonzero() indices."""
    rows = (mask != 0).any(dim=1)
    cols = (mask != 0).any(dim=0)
    H, W = mask.shape
    ar_h = torch.arange(H, device=mask.device)
    ar_w = torch.arange(W, device=mask.device)
    big = max(H, W) + 1
    ext = torch.stack([torch.where(rows, ar_h, big).min(), torch.where(rows, ar_h, -1).max(),
                       torch.where(cols, ar_w, big).min(), torch.where(cols, ar_w, -1).max()]).tolist()
    min_h, max_h, min_w, max_w = (int(v) for v in ext)
    if max_h < 0:
        raise IndexError("get_nonzero_region_tuple: empty mask")   # the reference's .min() on an empty tensor raises too
    size = max(max_h - min_h + 1, max_w - min_w + 1) * 1.1
    h_start = min(min_h, max_h) - (size - (max_h - min_h + 1)) / 2
    w_start = min(min_w, max_w) - (size - (max_w - min_w + 1)) / 2
    min_h = max(0, int(h_start))
    min_w = max(0, int(w_start))
    max_h = min(mask.shape[0], int(min_h + size))
    max_w = min(mask.shape[1], int(min_w + size))
    return min_h, min_w, max_h, max_w


def split_3x2_grid_to_tensor_with_6_elements(grid_image, tile_size):
    """[1,C,3t,2t] -> [6,C,t,t], tile index = 3*col + row (utils.py:347-371); stays on the input's device."""
    num_rows = grid_image.shape[2] // tile_size
    num_cols = grid_image.shape[3] // tile_size
    x = grid_image[0].reshape(grid_image.shape[1], num_rows, tile_size, num_cols, tile_size)
    return x.permute(3, 1, 0, 2, 4).reshape(num_rows * num_cols, grid_image.shape[1], tile_size, tile_size)


def merge_tensor_with_6_elements_to_3x2_grid(components, tile_size):
    num_rows, num_cols = 3, 2
    Cc = components.shape[1]
    x = components.reshape(num_cols, num_rows, Cc, tile_size, tile_size).permute(2, 1, 3, 0, 4)
    return x.reshape(1, Cc, num_rows * tile_size, num_cols * tile_size)


def scale_latents(latents):
    return (latents - 0.22) * 0.75


def unscale_latents(latents):
    return latents / 0.75 + 0.22


def scale_image(image):
    return image * 0.5 / 0.8


def unscale_image(image):
    return image / 0.5 * 0.8


class DreamTimeScheduler:
    """Time-prioritised SDS timestep schedule (src/training/trainer.py:54-106)."""
    def __init__(self, alphas_cumprod, total_iterations, m=750, s=125):
        self.total_iterations = total_iterations
        self.T = len(alphas_cumprod)
        w_d = torch.sqrt(1 - alphas_cumprod)
        timesteps = torch.arange(self.T, device=alphas_cumprod.device)
        w_p = torch.exp(-((timesteps - m) ** 2) / (2 * (s ** 2)))
        weights = w_d * w_p
        weights = weights / weights.sum()
        self.cumulative_survival = torch.flip(torch.cumsum(torch.flip(weights, dims=[0]), dim=0), dims=[0])

    def get_t(self, i):
        diffs = torch.abs(self.cumulative_survival - i / self.total_iterations)
        return torch.argmin(diffs).item()
