"""Terrain generation for `mesh_type` 'heightfield' / 'trimesh' (host side, numpy; runs once at env creation).

Mirrors the reference surface:
  * `HumanoidTerrain(cfg.terrain, num_envs)` -- humanoid/utils/terrain.py:189-234 on top of `Terrain` (:37-165):
    attributes `height_field_raw / heightsamples [tot_rows, tot_cols] int16`, `env_origins [rows, cols, 3]`,
    `border`, `tot_rows`, `tot_cols`, `env_length`, `env_width`, `vertices`, `triangles` (trimesh only).
  * the `isaacgym.terrain_utils` names the reference imports (`SubTerrain`, `random_uniform_terrain`,
    `pyramid_sloped_terrain`, `discrete_obstacles_terrain`, `pyramid_stairs_terrain`, `stepping_stones_terrain`,
    `convert_heightfield_to_trimesh`), restated from that module's published behaviour -- it is not part of the
    reference tree.  Random draws come from numpy's global generator in the same order, so a seeded run
    (`set_seed` seeds numpy, humanoid/utils/helpers.py:87-98) lays out the same map.

The device consumes the int16 grid through `hx_sim_set_terrain` (include/hx_sim.h); collision uses the two
triangles per cell of `convert_heightfield_to_trimesh` without the slope-threshold vertex shift (DESIGN.md
"Terrain").
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _upsample_linear(coarse, rows, cols):
    """Separable linear interpolation of coarse[a, b] (nodes on linspace(0,1,a) x linspace(0,1,b))."""
    a, b = coarse.shape
    ua, ub = np.linspace(0.0, 1.0, a), np.linspace(0.0, 1.0, b)
    tmp = np.stack([np.interp(np.linspace(0.0, 1.0, cols), ub, coarse[k]) for k in range(a)])      # [a, cols]
    return np.stack([np.interp(np.linspace(0.0, 1.0, rows), ua, tmp[:, c]) for c in range(cols)], 1)


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    vs, hs = terrain.vertical_scale, terrain.horizontal_scale
    lo, hi, inc = int(min_height / vs), int(max_height / vs), int(step / vs)
    shape = (int(terrain.width * hs / downsampled_scale), int(terrain.length * hs / downsampled_scale))
    coarse = np.random.choice(np.arange(lo, hi + inc, inc), shape)
    terrain.height_field_raw += np.rint(_upsample_linear(coarse.astype(np.float64), terrain.width, terrain.length)).astype(np.int16)
    return terrain


def sloped_terrain(terrain, slope=1):
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    ramp = (np.arange(terrain.width) / terrain.width).reshape(terrain.width, 1)
    terrain.height_field_raw[:, :] += (peak * ramp).astype(terrain.height_field_raw.dtype)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    tent_x = ((cx - np.abs(cx - np.arange(terrain.width))) / cx).reshape(terrain.width, 1)
    tent_y = ((cy - np.abs(cy - np.arange(terrain.length))) / cy).reshape(1, terrain.length)
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (peak * tent_x * tent_y).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    ref = terrain.height_field_raw[terrain.width // 2 - half, terrain.length // 2 - half]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(ref, 0), max(ref, 0))
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    hs = terrain.horizontal_scale
    h = int(max_height / terrain.vertical_scale)
    lo, hi, plat = int(min_size / hs), int(max_size / hs), int(platform_size / hs)
    rows, cols = terrain.height_field_raw.shape
    heights = [-h, -h // 2, h // 2, h]
    extents = range(lo, hi, 4)
    for _ in range(num_rects):
        w = np.random.choice(extents)
        ln = np.random.choice(extents)
        i = np.random.choice(range(0, rows - w, 4))
        j = np.random.choice(range(0, cols - ln, 4))
        terrain.height_field_raw[i:i + w, j:j + ln] = np.random.choice(heights)
    terrain.height_field_raw[(terrain.width - plat) // 2:(terrain.width + plat) // 2,
                             (terrain.length - plat) // 2:(terrain.length + plat) // 2] = 0
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    k = 0
    while (terrain.width - 2 * k * sw) > plat and (terrain.length - 2 * k * sw) > plat:
        k += 1
        terrain.height_field_raw[k * sw:terrain.width - k * sw, k * sw:terrain.length - k * sw] = k * sh
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    hs, vs = terrain.horizontal_scale, terrain.vertical_scale
    size, gap, hmax, plat = int(stone_size / hs), int(stone_distance / hs), int(max_height / vs), int(platform_size / hs)
    heights = np.arange(-hmax - 1, hmax, step=1)
    hf = terrain.height_field_raw
    hf[:, :] = int(depth / vs)
    W, Ln = terrain.width, terrain.length
    if Ln >= W:
        y = 0
        while y < Ln:
            y_end = min(Ln, y + size)
            x = np.random.randint(0, size)
            hf[0:max(0, x - gap), y:y_end] = np.random.choice(heights)
            while x < W:
                hf[x:min(W, x + size), y:y_end] = np.random.choice(heights)
                x += size + gap
            y += size + gap
    else:
        x = 0
        while x < W:
            x_end = min(W, x + size)
            y = np.random.randint(0, size)
            hf[x:x_end, 0:max(0, y - gap)] = np.random.choice(heights)
            while y < Ln:
                hf[x:x_end, y:min(Ln, y + size)] = np.random.choice(heights)
                y += size + gap
            x += size + gap
    hf[(W - plat) // 2:(W + plat) // 2, (Ln - plat) // 2:(Ln + plat) // 2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """(vertices [R*C,3] float32, triangles [2(R-1)(C-1),3] uint32).  Cell (i,j) -> (v00,v11,v01), (v00,v10,v11)."""
    hf = np.asarray(height_field_raw)
    R, Cn = hf.shape
    gx = np.repeat(np.linspace(0, (R - 1) * horizontal_scale, R)[:, None], Cn, 1)
    gy = np.repeat(np.linspace(0, (Cn - 1) * horizontal_scale, Cn)[None, :], R, 0)
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        d_i = hf[1:, :].astype(np.int64) - hf[:-1, :]
        d_j = hf[:, 1:].astype(np.int64) - hf[:, :-1]
        d_c = hf[1:, 1:].astype(np.int64) - hf[:-1, :-1]
        sx, sy, sc = np.zeros((R, Cn)), np.zeros((R, Cn)), np.zeros((R, Cn))
        sx[:-1, :] += d_i > thr
        sx[1:, :] -= -d_i > thr
        sy[:, :-1] += d_j > thr
        sy[:, 1:] -= -d_j > thr
        sc[:-1, :-1] += d_c > thr
        sc[1:, 1:] -= -d_c > thr
        gx = gx + (sx + sc * (sx == 0)) * horizontal_scale
        gy = gy + (sy + sc * (sy == 0)) * horizontal_scale
    vertices = np.stack([gx.ravel(), gy.ravel(), hf.ravel() * vertical_scale], 1).astype(np.float32)
    v00 = (np.arange(R - 1)[:, None] * Cn + np.arange(Cn - 1)[None, :]).ravel()
    tri = np.empty((v00.size, 2, 3), np.uint32)
    tri[:, 0, 0], tri[:, 0, 1], tri[:, 0, 2] = v00, v00 + Cn + 1, v00 + 1
    tri[:, 1, 0], tri[:, 1, 1], tri[:, 1, 2] = v00, v00 + Cn, v00 + Cn + 1
    return vertices, tri.reshape(-1, 3)


class Terrain:
    """Grid of `num_rows x num_cols` tiles inside a flat border (reference humanoid/utils/terrain.py:37-165).
    Sub-classes choose the tiles in `make_terrain`."""

    def __init__(self, cfg, num_robots):
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self.curiculum()
        elif getattr(cfg, "selected", False):
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw
        if self.type == "trimesh":
            self._mesh = None          # built on first access: 4.4 M vertices are only needed by viewers/exporters

    @property
    def vertices(self):
        return self._trimesh()[0]

    @property
    def triangles(self):
        return self._trimesh()[1]

    def _trimesh(self):
        if getattr(self, "_mesh", None) is None:
            self._mesh = convert_heightfield_to_trimesh(self.height_field_raw, self.cfg.horizontal_scale,
                                                        self.cfg.vertical_scale, self.cfg.slope_treshold)
        return self._mesh

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                self.add_terrain_to_map(self.make_terrain(j / self.cfg.num_cols + 0.001, i / self.cfg.num_rows), i, j)

    def selected_terrain(self):
        raise NotImplementedError("terrain.selected=True: the reference branch (humanoid/utils/terrain.py:92-105) "
                                  "dereferences attributes that are never set; no task uses it")

    def _tile(self):
        return SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                          vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)

    def make_terrain(self, choice, difficulty):
        raise NotImplementedError

    def add_terrain_to_map(self, terrain, row, col):
        x0 = self.border + row * self.length_per_env_pixels
        y0 = self.border + col * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = terrain.height_field_raw
        hs = terrain.horizontal_scale
        xa, xb = int((self.env_length / 2.0 - 1) / hs), int((self.env_length / 2.0 + 1) / hs)
        ya, yb = int((self.env_width / 2.0 - 1) / hs), int((self.env_width / 2.0 + 1) / hs)
        z = np.max(terrain.height_field_raw[xa:xb, ya:yb]) * terrain.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]


class HumanoidTerrain(Terrain):
    """Tile mix of the hector / humanoid tasks (reference humanoid/utils/terrain.py:189-234): by cumulative
    `terrain_proportions` -> flat, discrete obstacles, random uniform, slope up, slope down, stairs up, stairs down."""

    def __init__(self, cfg, num_robots):
        self.tile_log = []          # (kind, difficulty) per generated tile, in generation order
        super().__init__(cfg, num_robots)

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.uniform(0, 1)
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def make_terrain(self, choice, difficulty):
        t = self._tile()
        block_h, rough_h, slope = difficulty * 0.2, difficulty * 0.14, difficulty * 0.45
        kind = int(np.searchsorted(np.asarray(self.proportions, np.float64), choice, side="right"))
        self.tile_log.append((kind, float(difficulty)))       # diagnostics: (kind 0..6 as in the class docstring, difficulty)
        if kind == 1:
            discrete_obstacles_terrain(t, block_h, 1.0, 2.0, 20, platform_size=3.0)
        elif kind == 2:
            random_uniform_terrain(t, min_height=-rough_h, max_height=rough_h, step=0.005, downsampled_scale=0.2)
        elif kind == 3:
            pyramid_sloped_terrain(t, slope=slope, platform_size=0.1)
        elif kind == 4:
            pyramid_sloped_terrain(t, slope=-slope, platform_size=0.1)
        elif kind == 5:
            pyramid_stairs_terrain(t, step_width=0.4, step_height=block_h, platform_size=1.0)
        elif kind == 6:
            pyramid_stairs_terrain(t, step_width=0.4, step_height=-block_h, platform_size=1.0)
        return t
