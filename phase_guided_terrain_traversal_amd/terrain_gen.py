"""Terrain authoring on the host (N3 of SURVEY 8f): the box-stair tile set, the 14-tile adjacency rules and a
wave-function-collapse solver producing the `(T, 100, 10)` terrain tables the simulator consumes
(`[pos xyz, quat wxyz, half-size xyz]` per box; unused rows parked at 100+k like the shipped level files).

Restated from the reference's offline tools — runs once on the host, never on the hot path:
    terrain/generator.py:90-283   TerrainGenerator (AddBox / AddStairs / AddFlat / AddTurningStairsUp / Down)
    terrain/generator.py:288-326  generate_14: 5x5 map, border = tile 1 (platform), centre = tile 0 (bare floor)
    terrain/generator.py:328-358  addElement: tile index -> geometry
    terrain/generator.py:368-391  create_random_matrix (height ~ U, width ~ U(0.3, 0.45), steps in {2,3,4}; boxes beyond
                                  num_bodies are silently dropped)
    terrain/getIndexes.py:28-79   adjacency rules of the stair tiles
    wfc/wfc/wfc.py:42-229         WFCCore (here: an own min-entropy collapse + arc-consistency + backtracking solver)
Pinned by tests/golden/terrain_gen.npz (tile geometry and the rule table exactly; reference-solved waves must pass
this module's adjacency checker, and so must the maps produced here).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

N_TILES = 14
DIRECTIONS = [(-1, 0), (0, -1), (1, 0), (0, 1)]           # left, down, right, up (index offsets on the map array)


# ------------------------------------------------------------------ adjacency rules
def _rot_index(base: int, yaw: float) -> int:
    k = int(round((yaw % (2 * np.pi)) / (np.pi / 2))) % 4
    return base + k


def _s(yaw):  return _rot_index(2, yaw)       # straight stairs 2..5
def _su(yaw): return _rot_index(6, yaw)       # turning stairs up 6..9
def _sd(yaw): return _rot_index(10, yaw)      # turning stairs down 10..13


def connections() -> Dict[int, Dict[Tuple[int, int], Tuple[int, ...]]]:
    """allowed neighbour tiles: rules[tile][direction] (direction = index offset of the neighbour)."""
    left, down, right, up = DIRECTIONS
    rules: Dict[int, Dict[Tuple[int, int], Tuple[int, ...]]] = {
        0: {left: (0, 4, 10, 11), down: (0, 5, 11, 12), right: (0, 2, 12, 13), up: (0, 3, 13, 10)},
        1: {left: (1, 2, 6, 7), down: (1, 3, 7, 8), right: (1, 4, 8, 9), up: (1, 5, 9, 6)},
    }
    h = np.pi / 2
    for i in range(4):
        yaw = i * h
        d = [DIRECTIONS[(k + i) % 4] for k in range(4)]
        rules[_s(yaw)] = {d[0]: (_s(yaw + np.pi), 0), d[1]: (_s(yaw), _su(yaw), _sd(yaw - h)),
                          d[2]: (_s(yaw + np.pi), 1), d[3]: (_su(yaw + h), _s(yaw), _sd(yaw + np.pi))}
        rules[_su(yaw)] = {d[0]: (_su(yaw - h), _s(yaw - h), _sd(yaw + np.pi)), d[1]: (_s(yaw + h), 1),
                           d[2]: (_s(yaw + np.pi), 1), d[3]: (_su(yaw + h), _s(yaw), _sd(yaw + np.pi))}
        rules[_sd(yaw)] = {d[0]: (_s(yaw + h), _su(yaw + np.pi), _sd(yaw - h)), d[1]: (0, _s(yaw - h)),
                           d[2]: (_s(yaw), 0), d[3]: (_su(yaw + np.pi), _s(yaw + np.pi), _sd(yaw + h))}
    return rules


def rules_array(rules=None) -> np.ndarray:
    rules = rules or connections()
    a = np.zeros((N_TILES, 4, N_TILES), dtype=bool)
    for t in range(N_TILES):
        for di, d in enumerate(DIRECTIONS):
            a[t, di, list(rules[t][d])] = True
    return a


def check_wave(wave: np.ndarray, allowed: Optional[np.ndarray] = None, both: bool = False) -> bool:
    """Every adjacent pair satisfies the rule from at least one side (the reference only propagates from the cell
    collapsed first) — or from both sides when `both`."""
    allowed = rules_array() if allowed is None else allowed
    H, W = wave.shape
    for i in range(H):
        for j in range(W):
            for di, (a, b) in enumerate(DIRECTIONS):
                ni, nj = i + a, j + b
                if not (0 <= ni < H and 0 <= nj < W):
                    continue
                fwd = allowed[wave[i, j], di, wave[ni, nj]]
                back = allowed[wave[ni, nj], (di + 2) % 4, wave[i, j]]
                if (both and not (fwd and back)) or (not both and not (fwd or back)):
                    return False
    return True


# ------------------------------------------------------------------ wave function collapse
def solve_wfc(size: int, rng: np.random.Generator, fixed: Optional[Dict[Tuple[int, int], int]] = None,
              max_backtracks: int = 10000) -> np.ndarray:
    """Min-entropy collapse with backtracking.  Like the reference solver, a collapsed cell restricts its (still open)
    neighbours to the tiles its rule allows in that direction; adjacent pairs therefore satisfy the rule as seen from
    the cell that was collapsed first (`check_wave` semantics)."""
    allowed = rules_array()
    budget = [max_backtracks]

    def collapse(valid, done, wave, i, j, t):
        valid[i, j] = False; valid[i, j, t] = True
        done[i, j] = True; wave[i, j] = t
        for di, (a, b) in enumerate(DIRECTIONS):
            ni, nj = i + a, j + b
            if 0 <= ni < size and 0 <= nj < size and not done[ni, nj]:
                valid[ni, nj] &= allowed[t, di]
                if not valid[ni, nj].any():
                    return False
        return True

    valid0 = np.ones((size, size, N_TILES), dtype=bool)
    done0 = np.zeros((size, size), dtype=bool)
    wave0 = np.zeros((size, size), dtype=np.int32)
    for (i, j), t in (fixed or {}).items():
        if not valid0[i, j, t] or not collapse(valid0, done0, wave0, i, j, t):
            raise ValueError("fixed tiles are inconsistent with the adjacency rules")

    def search(valid, done, wave):
        if done.all():
            return wave
        counts = np.where(done, N_TILES + 1, valid.sum(-1))
        cand = np.argwhere(counts == counts.min())
        i, j = cand[rng.integers(len(cand))]
        for t in rng.permutation(np.flatnonzero(valid[i, j])):
            v, d, w = valid.copy(), done.copy(), wave.copy()
            if collapse(v, d, w, i, j, int(t)):
                out = search(v, d, w)
                if out is not None:
                    return out
            budget[0] -= 1
            if budget[0] <= 0:
                raise RuntimeError("WFC: backtracking budget exhausted")
        return None

    out = search(valid0, done0, wave0)
    if out is None:
        raise RuntimeError("WFC: no solution")
    return out


def generate_14(size: int, rng: np.random.Generator) -> np.ndarray:
    """size x size map: border = tile 1 (platform of full block height), centre = tile 0 (bare floor)."""
    fixed = {}
    for x in range(size):
        fixed[(x, 0)] = 1; fixed[(x, size - 1)] = 1
    for y in range(1, size - 1):
        fixed[(0, y)] = 1; fixed[(size - 1, y)] = 1
    fixed[(size // 2, size // 2)] = 0
    return solve_wfc(size, rng, fixed)


# ------------------------------------------------------------------ tile geometry
def _quat_yaw(yaw: float) -> np.ndarray:
    cz, sz = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.array([1.0 * 1.0 * cz + 0.0 * 0.0 * sz, 0.0 * 1.0 * cz - 1.0 * 0.0 * sz, 1.0 * 0.0 * cz + 0.0 * 1.0 * sz, 1.0 * 1.0 * sz - 0.0 * 0.0 * cz])


def _rot2d(x, y, yaw):
    return x * np.cos(yaw) - y * np.sin(yaw), x * np.sin(yaw) + y * np.cos(yaw)


class TileSet:
    def __init__(self, width: float, step_height: float, num_stairs: int):
        self.width, self.step_height, self.num_stairs = width, step_height, int(num_stairs)
        self.length = num_stairs * width
        self.block_height = num_stairs * step_height
        self.boxes: List[np.ndarray] = []

    def add_box(self, pos, yaw, size):
        self.boxes.append(np.concatenate([np.asarray(pos, dtype=np.float64), _quat_yaw(yaw), 0.5 * np.asarray(size, dtype=np.float64)]))

    def stairs(self, p, yaw):
        w, h, ns, L = self.width, self.step_height, self.num_stairs, self.length
        lx, lz = -w / 2, 0.0
        for _ in range(ns):
            lx += w; lz += h
            x, y = _rot2d(lx - ns * w / 2, L / 2 - ns * w / 2, yaw)
            self.add_box([x + p[0], y + p[1], lz / 2 + 0.0], yaw, [w, L, lz])

    def flat(self, p, height, width=0.1):
        L = self.length
        if height > 0.0:
            self.add_box([p[0], p[1], height / 2], 0.0, [L, L, height])
        else:
            self.add_box([p[0], p[1], height - width / 2], 0.0, [L, L, width])

    def _turning(self, p, yaw, up: bool):
        w, h, ns = self.width, self.step_height, self.num_stairs
        lp = [-ns * w / 2 - w / 2, ns * w / 2, 0.0 if up else ns * h + h]
        for i in range(ns):
            lp[0] += w; lp[1] -= w / 2; lp[2] += h if up else -h
            x, y = _rot2d(lp[0], lp[1], yaw)
            self.add_box([x + p[0], y + p[1], lp[2] / 2 + 0.0], yaw, [w, w + w * i, lp[2]])
        lp = [ns * w / 2 - w / 2, ns * w / 2, h if up else (ns - 1) * h + h]
        for i in range(ns - 1):
            lp[0] -= w; lp[1] -= w / 2; lp[2] += h if up else -h
            x, y = _rot2d(lp[0], lp[1], yaw + np.pi / 2)
            self.add_box([x + p[0], y + p[1], lp[2] / 2 + 0.0], yaw + np.pi / 2, [w, w + w * i, lp[2]])

    def add_tile(self, index: int, p) -> None:
        half = np.pi / 2
        if index == 0:
            return                                           # bare floor: the plane
        if index == 1:
            self.flat(p, self.block_height)
        elif 2 <= index <= 5:
            self.stairs(p, [0.0, half, np.pi, -half][index - 2])
        elif 6 <= index <= 9:
            self._turning(p, [0.0, half, np.pi, -half][index - 6], up=True)
        elif 10 <= index <= 13:
            self._turning(p, [0.0, half, np.pi, -half][index - 10], up=False)
        else:
            raise ValueError(index)


def centered_grid(n: int, d: float) -> np.ndarray:
    x = (np.arange(n) - (n - 1) / 2) * d
    X, Y = np.meshgrid(x, x, indexing="ij")
    return np.stack((X, Y), axis=-1)


def create_random_matrix(num_envs: int, num_bodies: int = 100, size: int = 5, height_min: float = 0.05, height_max: float = 0.13,
                         seed: int = 0) -> np.ndarray:
    """(num_envs, num_bodies, 10) float32 terrain table, one WFC map per variant (reference generator.py:368-391)."""
    rng = np.random.default_rng(seed)
    out = np.ones((num_envs, num_bodies, 10), dtype=np.float64)
    out[..., :3] = np.arange(100, 100 + num_envs * num_bodies).reshape(num_envs, num_bodies, 1)
    out[..., 3:7] = [1, 0, 0, 0]
    for e in range(num_envs):
        ts = TileSet(width=rng.uniform(0.3, 0.45), step_height=rng.uniform(height_min, height_max), num_stairs=int(rng.choice([2, 3, 4])))
        wave = generate_14(size, rng)
        grid = centered_grid(size, ts.length)
        for i in range(size):
            for j in range(size):
                ts.add_tile(int(wave[i, j]), grid[i, j])
        for i, row in enumerate(ts.boxes[:num_bodies]):          # boxes beyond num_bodies are dropped (reference quirk B13)
            out[e, i] = row
    return out.astype(np.float32)
