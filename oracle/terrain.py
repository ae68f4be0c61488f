"""CPU restatement of the terrain the hector task trains on (TEST INFRASTRUCTURE -- only tests/, smoke() and
bench.py's cpu_baseline leg may import this; the product never does).

Two layers:

* Tile assembly -- follows the reference file humanoid/utils/terrain.py line by line in behaviour
  (`Terrain.__init__` :37-71, `add_terrain_to_map` :148-165, `HumanoidTerrain.randomized_terrain` :194-203 and
  `make_terrain` :205-234, `Terrain.curiculum` :83-90).  PINNED: tests/golden/terrain_small.npz is produced by the
  reference's own `HumanoidTerrain` class running over these primitives (tests/golden/make_terrain_fixtures.py).
* Sub-terrain primitives -- the reference calls `isaacgym.terrain_utils` (NVIDIA Isaac Gym preview 4, not under
  /root/reference, closed distribution).  They are restated here from the published behaviour of that module
  (height grids in int16 units of `vertical_scale`, numpy global generator, the draw order below).
  PARITY UNPINNED at this boundary: the reference holds no test or fixture for them.

Plus the continuous height function used by the physics oracle: every grid cell is split into two triangles along
the (i,j)-(i+1,j+1) diagonal -- the split `convert_heightfield_to_trimesh` produces -- and a query returns the
height and unit normal of the triangle under (x, y).  The slope-threshold vertex shift that the reference applies
for `mesh_type='trimesh'` (legged_robot_config.py `slope_treshold`) turns steep grid steps into vertical walls:
`heightfield_to_trimesh` produces the shifted mesh (pinned by the fixture), and `HeightField(wall_height=...)` collides
with it the way the HIP kernel does (`contact()`: cliff cells continue at their low level, a point that entered a
plateau through a wall is pushed back horizontally; isaac_amd/csrc/hx_dyn.h terrain_query / wall_push); see DESIGN.md 4.1.
"""
import numpy as np


class SubTerrain:
    """One tile: `height_field_raw[width, length]` int16 in units of vertical_scale."""

    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((width, length), dtype=np.int16)


# ------------------------------------------------------------------------------------------ primitives
def _bilinear_resample(coarse, n_out_rows, n_out_cols):
    """coarse[a, b] sampled on linspace(0, 1, a) x linspace(0, 1, b), evaluated on linspace(0,1,n_out_*)."""
    a, b = coarse.shape
    out = np.zeros((n_out_rows, n_out_cols))
    for r in range(n_out_rows):
        u = r / (n_out_rows - 1) * (a - 1)
        i0 = min(int(np.floor(u)), a - 2)
        fu = u - i0
        for c in range(n_out_cols):
            v = c / (n_out_cols - 1) * (b - 1)
            j0 = min(int(np.floor(v)), b - 2)
            fv = v - j0
            out[r, c] = ((1 - fu) * (1 - fv) * coarse[i0, j0] + fu * (1 - fv) * coarse[i0 + 1, j0]
                         + (1 - fu) * fv * coarse[i0, j0 + 1] + fu * fv * coarse[i0 + 1, j0 + 1])
    return out


def random_uniform_terrain(t, min_height, max_height, step=1, downsampled_scale=None):
    """Heights drawn from a ladder of `step`-spaced levels on a coarse grid, bilinearly upsampled, rounded."""
    if downsampled_scale is None:
        downsampled_scale = t.horizontal_scale
    lo = int(min_height / t.vertical_scale)
    hi = int(max_height / t.vertical_scale)
    st = int(step / t.vertical_scale)
    ladder = np.arange(lo, hi + st, st)
    coarse = np.random.choice(ladder, (int(t.width * t.horizontal_scale / downsampled_scale),
                                       int(t.length * t.horizontal_scale / downsampled_scale)))
    fine = np.rint(_bilinear_resample(coarse.astype(np.float64), t.width, t.length))
    t.height_field_raw += fine.astype(np.int16)
    return t


def pyramid_sloped_terrain(t, slope=1, platform_size=1.0):
    """Pyramid: height = peak * tent(x) * tent(y), flattened at the height of the platform corner."""
    cx, cy = int(t.width / 2), int(t.length / 2)
    peak = int(slope * (t.horizontal_scale / t.vertical_scale) * (t.width / 2))
    add = np.zeros((t.width, t.length))
    for i in range(t.width):
        tx = (cx - abs(cx - i)) / cx
        for j in range(t.length):
            ty = (cy - abs(cy - j)) / cy
            add[i, j] = peak * tx * ty
    t.height_field_raw += add.astype(t.height_field_raw.dtype)
    half = int(platform_size / t.horizontal_scale / 2)
    x1, y1 = t.width // 2 - half, t.length // 2 - half
    corner = t.height_field_raw[x1, y1]
    t.height_field_raw = np.clip(t.height_field_raw, min(corner, 0), max(corner, 0))
    return t


def discrete_obstacles_terrain(t, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """`num_rects` axis-aligned blocks of height in {-h, -h/2, h/2, h}; the centre platform is cleared."""
    h = int(max_height / t.vertical_scale)
    smin = int(min_size / t.horizontal_scale)
    smax = int(max_size / t.horizontal_scale)
    plat = int(platform_size / t.horizontal_scale)
    rows, cols = t.height_field_raw.shape
    levels = [-h, -h // 2, h // 2, h]
    sizes = range(smin, smax, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes)
        ln = np.random.choice(sizes)
        i0 = np.random.choice(range(0, rows - w, 4))
        j0 = np.random.choice(range(0, cols - ln, 4))
        t.height_field_raw[i0:i0 + w, j0:j0 + ln] = np.random.choice(levels)
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    t.height_field_raw[x1:x2, y1:y2] = 0
    return t


def pyramid_stairs_terrain(t, step_width, step_height, platform_size=1.0):
    """Concentric square steps rising (or descending) towards the centre platform."""
    sw = int(step_width / t.horizontal_scale)
    sh = int(step_height / t.vertical_scale)
    plat = int(platform_size / t.horizontal_scale)
    level = 0
    x0, x1, y0, y1 = 0, t.width, 0, t.length
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw
        x1 -= sw
        y0 += sw
        y1 -= sw
        level += sh
        t.height_field_raw[x0:x1, y0:y1] = level
    return t


def stepping_stones_terrain(t, stone_size, stone_distance, max_height, platform_size=1.0, depth=-10):
    """Square stones at random heights separated by holes of `depth`; only reachable through the reference's
    base `Terrain.make_terrain` (not used by the hector task)."""
    size = int(stone_size / t.horizontal_scale)
    dist = int(stone_distance / t.horizontal_scale)
    hmax = int(max_height / t.vertical_scale)
    plat = int(platform_size / t.horizontal_scale)
    levels = np.arange(-hmax - 1, hmax, step=1)
    t.height_field_raw[:, :] = int(depth / t.vertical_scale)
    sx = sy = 0
    if t.length >= t.width:
        while sy < t.length:
            ey = min(t.length, sy + size)
            sx = np.random.randint(0, size)
            ex = max(0, sx - dist)
            t.height_field_raw[0:ex, sy:ey] = np.random.choice(levels)
            while sx < t.width:
                ex = min(t.width, sx + size)
                t.height_field_raw[sx:ex, sy:ey] = np.random.choice(levels)
                sx += size + dist
            sy += size + dist
    else:
        while sx < t.width:
            ex = min(t.width, sx + size)
            sy = np.random.randint(0, size)
            ey = max(0, sy - dist)
            t.height_field_raw[sx:ex, 0:ey] = np.random.choice(levels)
            while sy < t.length:
                ey = min(t.length, sy + size)
                t.height_field_raw[sx:ex, sy:ey] = np.random.choice(levels)
                sy += size + dist
            sx += size + dist
    x1, x2 = (t.width - plat) // 2, (t.width + plat) // 2
    y1, y2 = (t.length - plat) // 2, (t.length + plat) // 2
    t.height_field_raw[x1:x2, y1:y2] = 0
    return t


def heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Vertices [rows*cols, 3] float32 and triangles [2*(rows-1)*(cols-1), 3] uint32; each cell (i, j) gives
    (v00, v11, v01) and (v00, v10, v11).  With a slope threshold, the low vertex of an edge steeper than the
    threshold is pulled one cell towards the high one, so steep ramps become vertical walls."""
    hf = height_field_raw
    rows, cols = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (cols - 1) * horizontal_scale, cols),
                         np.linspace(0, (rows - 1) * horizontal_scale, rows))
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        mx = np.zeros((rows, cols))
        my = np.zeros((rows, cols))
        mc = np.zeros((rows, cols))
        mx[:rows - 1, :] += (hf[1:, :] - hf[:rows - 1, :] > thr)
        mx[1:, :] -= (hf[:rows - 1, :] - hf[1:, :] > thr)
        my[:, :cols - 1] += (hf[:, 1:] - hf[:, :cols - 1] > thr)
        my[:, 1:] -= (hf[:, :cols - 1] - hf[:, 1:] > thr)
        mc[:rows - 1, :cols - 1] += (hf[1:, 1:] - hf[:rows - 1, :cols - 1] > thr)
        mc[1:, 1:] -= (hf[:rows - 1, :cols - 1] - hf[1:, 1:] > thr)
        xx = xx + (mx + mc * (mx == 0)) * horizontal_scale
        yy = yy + (my + mc * (my == 0)) * horizontal_scale
    verts = np.zeros((rows * cols, 3), np.float32)
    verts[:, 0] = xx.flatten()
    verts[:, 1] = yy.flatten()
    verts[:, 2] = hf.flatten() * vertical_scale
    tris = -np.ones((2 * (rows - 1) * (cols - 1), 3), np.uint32)
    for i in range(rows - 1):
        v00 = np.arange(0, cols - 1) + i * cols
        v01, v10, v11 = v00 + 1, v00 + cols, v00 + cols + 1
        a, b = 2 * i * (cols - 1), 2 * i * (cols - 1) + 2 * (cols - 1)
        tris[a:b:2, 0], tris[a:b:2, 1], tris[a:b:2, 2] = v00, v11, v01
        tris[a + 1:b:2, 0], tris[a + 1:b:2, 1], tris[a + 1:b:2, 2] = v00, v10, v11
    return verts, tris


# ------------------------------------------------------------------------------------------ tile assembly
class HumanoidTerrainOracle:
    """reference humanoid/utils/terrain.py: Terrain.__init__ (:37-71) with HumanoidTerrain's overrides (:189-234)."""

    def __init__(self, cfg, num_robots):
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [float(np.sum(cfg.terrain_proportions[:i + 1])) for i in range(len(cfg.terrain_proportions))]
        self.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            for j in range(cfg.num_cols):                      # :83-90
                for i in range(cfg.num_rows):
                    self._place(self.make_terrain(j / cfg.num_cols + 0.001, i / cfg.num_rows), i, j)
        elif getattr(cfg, "selected", False):
            raise NotImplementedError("terrain.selected: the reference's own branch (:92-105) reads attributes "
                                      "that do not exist and cannot run")
        else:
            for k in range(self.num_sub_terrains):             # :194-203
                i, j = np.unravel_index(k, (cfg.num_rows, cfg.num_cols))
                choice = np.random.uniform(0, 1)
                difficulty = np.random.uniform(0, 1)
                self._place(self.make_terrain(choice, difficulty), i, j)
        self.heightsamples = self.height_field_raw

    def make_terrain(self, choice, difficulty):                # :205-234
        c = self.cfg
        t = SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                       vertical_scale=c.vertical_scale, horizontal_scale=c.horizontal_scale)
        obstacle_h = difficulty * 0.2
        rough_h = difficulty * 0.14
        slope = difficulty * 0.45
        p = self.proportions
        if choice < p[0]:
            pass
        elif choice < p[1]:
            discrete_obstacles_terrain(t, obstacle_h, 1.0, 2.0, 20, platform_size=3.0)
        elif choice < p[2]:
            random_uniform_terrain(t, min_height=-rough_h, max_height=rough_h, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            pyramid_sloped_terrain(t, slope=slope, platform_size=0.1)
        elif choice < p[4]:
            pyramid_sloped_terrain(t, slope=-slope, platform_size=0.1)
        elif choice < p[5]:
            pyramid_stairs_terrain(t, step_width=0.4, step_height=obstacle_h, platform_size=1.0)
        elif choice < p[6]:
            pyramid_stairs_terrain(t, step_width=0.4, step_height=-obstacle_h, platform_size=1.0)
        return t

    def _place(self, t, i, j):                                 # add_terrain_to_map :148-165
        x0 = self.border + i * self.length_per_env_pixels
        y0 = self.border + j * self.width_per_env_pixels
        self.height_field_raw[x0:x0 + self.length_per_env_pixels, y0:y0 + self.width_per_env_pixels] = t.height_field_raw
        x1 = int((self.env_length / 2.0 - 1) / t.horizontal_scale)
        x2 = int((self.env_length / 2.0 + 1) / t.horizontal_scale)
        y1 = int((self.env_width / 2.0 - 1) / t.horizontal_scale)
        y2 = int((self.env_width / 2.0 + 1) / t.horizontal_scale)
        z = np.max(t.height_field_raw[x1:x2, y1:y2]) * t.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]


# ------------------------------------------------------------------------------------------ height queries
class HeightField:
    """Continuous surface over the grid: world x = i*hs - border, y = j*hs - border (the transform the reference
    gives the mesh, legged_robot.py:561-562,578-579); outside the grid the border row/column continues."""

    def __init__(self, height_field_raw, horizontal_scale, vertical_scale, border_size, wall_height=0.0):
        # heights are held as float32 numbers, the precision of the mesh vertices the reference hands to the
        # simulator (convert_heightfield_to_trimesh returns float32) and of the device copy
        self.h = (np.asarray(height_field_raw, np.float64) * vertical_scale).astype(np.float32).astype(np.float64)
        self.hs = float(horizontal_scale)
        self.x0 = self.y0 = -float(border_size)
        # mesh_type 'trimesh': slope_treshold * horizontal_scale (utils/terrain.py:70-73, legged_robot_config.py:67).  Where
        # grid neighbours differ by more than this, convert_heightfield_to_trimesh moves the low vertex under the high one:
        # the low ground continues flat through the cell and a vertical wall stands on the high vertices' grid line.
        # 0 = 'heightfield' semantics (ramps).  PARITY UNPINNED like the rest of the contact model (isaacgym is absent).
        self.wall = float(wall_height)

    def query(self, x, y):
        """(height [n], unit normal [n,3]) of the triangle under each (x, y)."""
        x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
        rows, cols = self.h.shape
        u = (x - self.x0) / self.hs
        v = (y - self.y0) / self.hs
        i = np.clip(np.floor(u).astype(np.int64), 0, rows - 2)
        j = np.clip(np.floor(v).astype(np.int64), 0, cols - 2)
        fu = np.clip(u - i, 0.0, 1.0)
        fv = np.clip(v - j, 0.0, 1.0)
        h00, h10, h01, h11 = self.h[i, j], self.h[i + 1, j], self.h[i, j + 1], self.h[i + 1, j + 1]
        if self.wall > 0.0:
            # a cell that contains a wall: its "high" vertices (more than a wall height above the lowest one) have no
            # surface inside the cell, the low level takes their place
            lo = np.minimum(np.minimum(h00, h01), np.minimum(h10, h11))
            cliff = np.maximum(np.maximum(h00, h01), np.maximum(h10, h11)) - lo > self.wall
            h00, h01, h10, h11 = (np.where(cliff & (h - lo > self.wall), lo, h) for h in (h00, h01, h10, h11))
        upper = fv > fu                      # triangle (v00, v11, v01); otherwise (v00, v10, v11)
        gu = np.where(upper, h11 - h01, h10 - h00)
        gv = np.where(upper, h01 - h00, h11 - h10)
        z = h00 + fu * gu + fv * gv
        nrm = np.stack([-gu / self.hs, -gv / self.hs, np.ones_like(z)], -1)
        nrm /= np.sqrt(np.einsum("ni,ni->n", nrm, nrm))[:, None]
        return z, nrm

    def contact(self, x, y, z):
        """(penetration [n], unit normal [n,3]) of world points: distance to the plane of the triangle under the point, or
        -- for a point below a plateau's surface that is closer to one of the plateau's walls than to the surface above it
        -- the horizontal distance to that wall with the wall's outward normal (isaac_amd/csrc/hx_dyn.h wall_push)."""
        h, nrm = self.query(x, y)
        z = np.asarray(z, np.float64)
        pen = (h - z) * nrm[:, 2]
        if self.wall <= 0.0:
            return pen, nrm
        rows, cols = self.h.shape
        u = (np.asarray(x, np.float64) - self.x0) / self.hs
        v = (np.asarray(y, np.float64) - self.y0) / self.hs
        i = np.clip(np.floor(u).astype(np.int64), 1, rows - 3)
        j = np.clip(np.floor(v).astype(np.int64), 1, cols - 3)
        fu = np.clip(u - i, 0.0, 1.0)
        fv = np.clip(v - j, 0.0, 1.0)
        jn = j + (fv > 0.5)
        i_n = i + (fu > 0.5)
        g, W = self.h, self.wall
        best = pen.copy()
        dirn = np.full(len(pen), -1)
        for k, (top, low, d) in enumerate(((g[i, jn], g[i - 1, jn], fu * self.hs), (g[i + 1, jn], g[i + 2, jn], (1 - fu) * self.hs),
                                           (g[i_n, j], g[i_n, j - 1], fv * self.hs), (g[i_n, j + 1], g[i_n, j + 2], (1 - fv) * self.hs))):
            hit = (pen > 0) & (top - low > W) & (z < top) & (z > low - 0.5 * W) & (d < best)
            best = np.where(hit, d, best)
            dirn = np.where(hit, k, dirn)
        normals = np.array([[-1.0, 0, 0], [1.0, 0, 0], [0, -1.0, 0], [0, 1.0, 0]])
        hitany = dirn >= 0
        nrm = np.where(hitany[:, None], normals[np.maximum(dirn, 0)], nrm)
        return best, nrm
