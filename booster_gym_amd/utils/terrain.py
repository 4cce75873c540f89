"""Terrain: integer height field + bilinear height query.

Mirrors the interface of the reference `utils/terrain.py:6-121` (class `Terrain` with `type`,
`env_width`, `env_length`, `border_size`, `horizontal_scale`, `vertical_scale`, `border_pixels`,
`height_field_raw`, `terrain_heights(base_pos)`).  Differences by design:
  * no triangle mesh is produced: the HIP contact kernel collides sole corners with the height
    field directly (booster_gym_amd/csrc/bg_dyn.h: terrain_query), so `_create_trimesh`'s
    `convert_heightfield_to_trimesh` / `gym.add_triangle_mesh` (terrain.py:86-99) have no counterpart;
  * the sub-terrain generators (third-party `isaacgym.terrain_utils`, absent here) are restated from
    their documented behaviour with this module's own seeded numpy Generator;
  * `terrain_heights` on a CUDA tensor never leaves the device in the training path (the kernels
    query the field themselves); this method exists for set-up code and tests.
"""
import numpy as np
import torch


def _random_uniform(rng, width, length, hscale, vscale, min_height, max_height, step, downsampled_scale):
    lo, hi, st = int(min_height / vscale), int(max_height / vscale), max(int(step / vscale), 1)
    levels = np.arange(lo, hi + st, st)
    nx, ny = int(width * hscale / downsampled_scale), int(length * hscale / downsampled_scale)
    coarse = rng.choice(levels, size=(nx, ny)).astype(np.float64)
    # separable linear up-sampling of the coarse grid onto the pixel grid
    xs, ys = np.linspace(0.0, nx - 1.0, width), np.linspace(0.0, ny - 1.0, length)
    x0 = np.clip(np.floor(xs).astype(int), 0, nx - 2)
    y0 = np.clip(np.floor(ys).astype(int), 0, ny - 2)
    fx, fy = (xs - x0)[:, None], (ys - y0)[None, :]
    c00, c10 = coarse[x0][:, y0], coarse[x0 + 1][:, y0]
    c01, c11 = coarse[x0][:, y0 + 1], coarse[x0 + 1][:, y0 + 1]
    fine = (1 - fx) * (1 - fy) * c00 + fx * (1 - fy) * c10 + (1 - fx) * fy * c01 + fx * fy * c11
    return np.rint(fine).astype(np.int16)


def _discrete_obstacles(rng, width, length, hscale, vscale, max_height, min_size, max_size, num_rects, platform_size):
    hf = np.zeros((width, length), dtype=np.int16)
    mh = int(max_height / vscale)
    sizes = np.arange(int(min_size / hscale), int(max_size / hscale), 4)
    heights = np.array([-mh, -mh // 2, mh // 2, mh])
    for _ in range(num_rects):
        w, l = int(rng.choice(sizes)), int(rng.choice(sizes))
        i0 = int(rng.choice(np.arange(0, width - w, 4)))
        j0 = int(rng.choice(np.arange(0, length - l, 4)))
        hf[i0 : i0 + w, j0 : j0 + l] = rng.choice(heights)
    p = int(platform_size / hscale)
    x1, x2, y1, y2 = (width - p) // 2, (width + p) // 2, (length - p) // 2, (length + p) // 2
    hf[x1:x2, y1:y2] = 0
    return hf


def _pyramid_slope(width, length, hscale, vscale, slope, platform_size):
    cx, cy = int(width / 2), int(length / 2)
    xx = ((cx - np.abs(cx - np.arange(width))) / cx).reshape(width, 1)
    yy = ((cy - np.abs(cy - np.arange(length))) / cy).reshape(1, length)
    max_h = int(slope * (hscale / vscale) * (width / 2))
    hf = (max_h * xx * yy).astype(np.int16)
    p = int(platform_size / hscale / 2)
    ref = hf[width // 2 - p, length // 2 - p]
    return np.clip(hf, min(ref, 0), max(ref, 0)).astype(np.int16)


class Terrain:
    def __init__(self, device, terrain_cfg, seed=0):
        self.terrain_cfg = terrain_cfg
        self.device = device
        self.type = terrain_cfg["type"]
        if self.type == "plane":
            self.height_field_raw = None
        elif self.type == "trimesh":
            self._create_heightfield(seed)
        else:
            raise ValueError(f"Invalid terrain type: {self.type}")

    def _create_heightfield(self, seed):
        c = self.terrain_cfg
        self.env_width = c["num_terrains"] * c["terrain_width"]
        self.env_length = c["terrain_length"]
        self.border_size = c["border_size"]
        self.horizontal_scale = c["horizontal_scale"]
        self.vertical_scale = c["vertical_scale"]
        self.border_pixels = int(self.border_size / self.horizontal_scale)
        wpx, lpx = int(c["terrain_width"] / self.horizontal_scale), int(c["terrain_length"] / self.horizontal_scale)
        hf = np.zeros((c["num_terrains"] * wpx + 2 * self.border_pixels, lpx + 2 * self.border_pixels), dtype=np.int16)
        props = np.asarray(c["terrain_proportions"], dtype=np.float64)
        bounds = c["num_terrains"] * np.cumsum(props) / np.sum(props)
        rng = np.random.default_rng(seed)
        for i in range(c["num_terrains"]):
            if i < bounds[0]:
                sub = np.zeros((wpx, lpx), dtype=np.int16)
            elif i < bounds[1]:
                sub = _pyramid_slope(wpx, lpx, self.horizontal_scale, self.vertical_scale, c["slope"], 3.0)
            elif i < bounds[2]:
                sub = _random_uniform(rng, wpx, lpx, self.horizontal_scale, self.vertical_scale, -0.5 * c["random_height"],
                                      0.5 * c["random_height"], 0.005, 0.2)
            else:
                sub = _discrete_obstacles(rng, wpx, lpx, self.horizontal_scale, self.vertical_scale, c["discrete_height"], 1.0, 2.0, 20, 3.0)
            x0 = self.border_pixels + i * wpx
            hf[x0 : x0 + wpx, self.border_pixels : self.border_pixels + lpx] = sub
        self.height_field_raw = hf

    def terrain_heights(self, base_pos):
        """Bilinear height under world (x, y) -- reference terrain.py:101-121, indices clamped to the field."""
        if self.type == "plane":
            return torch.zeros(len(base_pos), dtype=torch.float, device=base_pos.device if torch.is_tensor(base_pos) else self.device)
        p = base_pos.detach().cpu().numpy() if torch.is_tensor(base_pos) else np.asarray(base_pos)
        x = self.border_pixels + p[:, 0] / self.horizontal_scale
        y = self.border_pixels + p[:, 1] / self.horizontal_scale
        hf = self.height_field_raw
        x1 = np.clip(np.floor(x).astype(int), 0, hf.shape[0] - 2)
        y1 = np.clip(np.floor(y).astype(int), 0, hf.shape[1] - 2)
        fx, fy = x - x1, y - y1
        h = ((1 - fx) * (1 - fy) * hf[x1, y1] + fx * (1 - fy) * hf[x1 + 1, y1] + (1 - fx) * fy * hf[x1, y1 + 1] + fx * fy * hf[x1 + 1, y1 + 1])
        dev = base_pos.device if torch.is_tensor(base_pos) else self.device
        return torch.tensor(h * self.vertical_scale, dtype=torch.float, device=dev)
