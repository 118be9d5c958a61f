"""Host-side mirror of the reference env interface (go2/joystick_pgtt.py:35-48, go2/base.py:216-231):

    env = Joystick(task="stairs", config=training_config(), num_envs=4096, terrain=level4, device="cuda:0")
    obs = env.reset(seed)                 # {'state': [N,171], 'privileged_state': [N,215]} torch tensors
    obs, reward, done, info = env.step(action)   # action [N,12] in [-1,1], FR,FL,RR,RL order

The reference env is functional (State pytrees under jax.vmap); this one is the batched, stateful
equivalent: state lives in caller-visible torch tensors (SoA in HBM) and every call enqueues HIP
kernels of libpgtt.so on the current torch stream.  PyTorch is only the allocator / stream provider.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import abi, configs, mjcf, native


class Joystick:
    """Batched PGTT joystick env on one GPU.  Properties mirror reference go2/base.py:216-231."""

    def __init__(self, task: str = "flat_terrain", config: Optional[Dict[str, Any]] = None, num_envs: int = 4096,
                 terrain: Optional[np.ndarray] = None, device: str = "cuda:0", params: Optional[torch.Tensor] = None,
                 variant: Optional[torch.Tensor] = None, box_friction: Optional[torch.Tensor] = None,
                 autoreset: bool = False, debug_contacts: bool = False, env_id_offset: int = 0,
                 model: Optional[Dict[str, Any]] = None, layout: Optional[str] = None, observe_form: Optional[str] = None,
                 test_hooks: bool = False, interval_sums: bool = False):
        """layout: "auto" | "quad" | "oct" | "hex" lane layout of physics_kernel (PgttConfig.lane_layout; results are bit-identical
        across batch sizes and shards within one layout); observe_form: "fused" | "split"; test_hooks: allow set_test_overrides
        (fixture replay only); interval_sums: keep per-env running sums of the step outputs for a logging trainer (PgttBuffers.interval_sums)."""
        self._config = dict(configs.default_config() if config is None else config)
        self._config["autoreset"] = int(autoreset)
        if layout is not None:
            self._config["lane_layout"] = layout
        if observe_form is not None:
            self._config["observe_form"] = observe_form
        if test_hooks:
            self._config["test_hooks"] = True
        self.method = self._config.get("method", "pgtt")     # "pgtt" = go2/joystick_pgtt.py, "baseline" = go2/joystick.py
        self.task = task
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise native.PgttError("Joystick needs a ROCm device (device='cuda:N'); there is no CPU path")
        self._model = mjcf.load_model(task) if model is None else model
        self._ms = abi.model_struct(self._model)
        self._cs = abi.config_struct(self._config)
        self._lib = native.lib()
        self._h = C.c_void_p()
        native.check(self._lib.pgtt_create(C.byref(self._cs), C.byref(self._ms), self.device.index or 0,
                                           self.num_envs, C.byref(self._h)))
        self.env_id_offset = int(env_id_offset)
        self.terrain = None
        if task == "stairs":
            if terrain is None:
                raise ValueError("task='stairs' needs a terrain table (T,B,10), e.g. assets/terrains/level4.npy")
            self.set_terrain(terrain)
        n = self.num_envs
        self.buffers: Dict[str, torch.Tensor] = {}
        for spec in abi.BUFFER_SPECS:
            dt = torch.float32 if spec[2] == np.float32 else torch.int32
            self.buffers[spec[0]] = torch.zeros(abi.buffer_shape(spec, n, self.method), dtype=dt, device=self.device)
        # the 22 metric rows, the reward row and the done row live in ONE [24][N] block, so that a trainer can reduce
        # them over the envs with a single kernel (distributed.MetricReducer.accumulate_block)
        self.step_block = torch.zeros((abi.NMETRIC + 2, n), dtype=torch.float32, device=self.device)
        self.buffers["metrics"] = self.step_block[:abi.NMETRIC]
        self.buffers["reward"] = self.step_block[abi.NMETRIC]
        self.buffers["done"] = self.step_block[abi.NMETRIC + 1]
        # per-env running sums of [22 metrics; reward; done] since the trainer last cleared them: a log interval then costs one
        # reduction over the envs (distributed.MetricReducer.reduce_block), not one per step
        # (opt-in: 24 read-modify-writes per env-step that only a logging loop such as bench.py consumes and clears)
        if interval_sums:
            self.buffers["interval_sums"] = torch.zeros((abi.NMETRIC + 2, n), dtype=torch.float32, device=self.device)
        if params is not None:
            self.buffers["params"] = params.to(self.device, torch.float32).contiguous()
            assert self.buffers["params"].shape == (abi.NPARAM, n)
        if variant is not None:
            self.buffers["variant"] = variant.to(self.device, torch.int32).contiguous()
        if box_friction is not None:
            self.buffers["box_friction"] = box_friction.to(self.device, torch.float32).contiguous()
            assert self.buffers["box_friction"].shape == (abi.MAX_BOX, n)
        if debug_contacts:
            self.buffers["dbg_contact"] = torch.zeros((n, abi.NCON * 2), dtype=torch.int32, device=self.device)
            self.buffers["dbg_dist"] = torch.zeros((n, abi.NCON), dtype=torch.float32, device=self.device)
            self.buffers["dbg_niter"] = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self._bind()
        self._seed = 0

    # ---- reference-compatible properties
    @property
    def dt(self) -> float:
        return self._config["ctrl_dt"]

    @property
    def action_size(self) -> int:
        return abi.NU

    @property
    def observation_size(self) -> Dict[str, int]:
        od, pd = abi.obs_dims(self.method)
        return {"state": od, "privileged_state": pd}

    @property
    def config(self) -> Dict[str, Any]:
        return self._config

    # go2/base.py:216-231 also exposes the compiled model and where it came from.  Here the "MuJoCo model" is the dict of compiled
    # constants (mjcf.compile_mjcf of go2_mjx_feetonly.xml + scene, shipped as assets/go2_<task>.json), and its device form is the
    # PgttModel struct the kernels read.
    @property
    def mj_model(self) -> Dict[str, Any]:
        return self._model

    @property
    def mjx_model(self) -> "abi.PgttModel":
        return self._ms

    @property
    def xml_path(self) -> str:
        return mjcf.asset_path(self.task)

    @property
    def model(self) -> Dict[str, Any]:
        return self._model

    def _bind(self) -> None:
        b = abi.PgttBuffers()
        for name, _ in abi.PgttBuffers._fields_:
            t = self.buffers.get(name)
            setattr(b, name, None if t is None else t.data_ptr())
        native.check(self._lib.pgtt_bind(self._h, C.byref(b)))

    def set_terrain(self, terrain: np.ndarray) -> None:
        t = np.ascontiguousarray(terrain, dtype=np.float32)
        assert t.ndim == 3 and t.shape[2] == 10 and t.shape[1] <= abi.MAX_BOX
        native.check(self._lib.pgtt_set_terrain(self._h, t.ctypes.data, t.shape[0], t.shape[1]))
        self.terrain = t

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _obs(self) -> Dict[str, torch.Tensor]:
        return {"state": self.buffers["obs_state"], "privileged_state": self.buffers["obs_priv"]}

    def reset(self, seed: int = 0, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        self._seed = int(seed)
        mp = None
        if mask is not None:
            mask = mask.to(self.device, torch.uint8).contiguous()
            mp = mask.data_ptr()
        native.check(self._lib.pgtt_reset(self._h, self._seed, self.env_id_offset, mp, self._stream()))
        return self._obs()

    def step(self, action: torch.Tensor):
        a = action.to(self.device, torch.float32).contiguous()
        assert a.shape == (self.num_envs, abi.NU)
        native.check(self._lib.pgtt_step(self._h, a.data_ptr(), self._stream()))
        info = {"metrics": self.buffers["metrics"], "episode_metrics": self.buffers["ep_metrics"]}
        return self._obs(), self.buffers["reward"], self.buffers["done"], info

    def physics(self, action: torch.Tensor) -> None:
        a = action.to(self.device, torch.float32).contiguous()
        native.check(self._lib.pgtt_physics(self._h, a.data_ptr(), self._stream()))

    def observe(self, action: torch.Tensor) -> None:
        a = action.to(self.device, torch.float32).contiguous()
        native.check(self._lib.pgtt_observe(self._h, a.data_ptr(), self._stream()))

    def scan(self, yaw: Optional[float] = None) -> torch.Tensor:
        native.check(self._lib.pgtt_scan(self._h, float("nan") if yaw is None else float(yaw), self._stream()))
        return self.buffers["scan_z"]

    def interval_reduce(self, out: torch.Tensor, env_steps: float = 0.0, accumulate: bool = False) -> None:
        """out[k] (+)= sum over the envs of buffers['interval_sums'][k] (k < NMETRIC + 2), out[NMETRIC + 2] (+)= env_steps, the rows cleared:
        one launch (pgtt_interval_reduce)"""
        if "interval_sums" not in self.buffers:
            raise native.PgttError("interval_reduce: this env keeps no interval sums - create it with Joystick(..., interval_sums=True)")
        if not (out.dtype == torch.float32 and out.is_contiguous() and out.numel() == abi.NMETRIC + 3 and out.device == self.buffers["interval_sums"].device):
            raise ValueError(f"interval_reduce: `out` must be {abi.NMETRIC + 3} contiguous float32 values on {self.buffers['interval_sums'].device}")
        native.check(self._lib.pgtt_interval_reduce(self._h, out.data_ptr(), float(env_steps), int(accumulate), self._stream()))

    def set_test_overrides(self, rng_value: Optional[float] = None, scan_preset: bool = False) -> None:
        """test hooks of libpgtt (include/pgtt.h): fixed uniform draws / scan heights taken from buffers['scan_z']"""
        native.check(self._lib.pgtt_set_test_overrides(self._h, float("nan") if rng_value is None else float(rng_value), int(scan_preset)))

    def enable_timing(self, on=True) -> None:
        """False / True / n > 1 = time every n-th step (HIP events around the kernels, on the launch stream)"""
        native.check(self._lib.pgtt_enable_timing(self._h, int(on)))

    def last_kernel_ms(self):
        p, o = C.c_float(), C.c_float()
        native.check(self._lib.pgtt_last_kernel_ms(self._h, C.byref(p), C.byref(o)))
        return p.value, o.value

    def kernel_ms_mean(self):
        """(physics ms, observe ms, steps): mean kernel times over all steps since enable_timing(True)"""
        p, o, k = C.c_float(), C.c_float(), C.c_int()
        native.check(self._lib.pgtt_kernel_ms_mean(self._h, C.byref(p), C.byref(o), C.byref(k)))
        return p.value, o.value, k.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.pgtt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
