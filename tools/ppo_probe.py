import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from phase_guided_terrain_traversal_amd import configs, ppo, abi
from phase_guided_terrain_traversal_amd.env import Joystick
env = Joystick("flat_terrain", configs.training_config(), num_envs=4096, device="cuda:0", autoreset=True)
env.reset(seed=0)
dev = env.device
model = ppo.ActorCritic().to(dev); ns = ppo.RunningNorm(abi.OBS, dev)
z = lambda: torch.zeros((), device=dev)
for ug in (True, False):
    a = ppo._Actor(env, model, ns, 40, ppo.PPOConfig(), 1000, acc=(z(), z(), z(), torch.zeros(abi.NMETRIC, device=dev)), use_graph=ug)
    print("graph active:", a.graph is not None)
    with torch.no_grad():
        a.rollout(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): a.rollout()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("use_graph", ug, "rollout of 40 steps:", round(dt * 1e3, 2), "ms ->", round(40 * 4096 / dt / 1e6, 2), "M steps/s")

# ---- the rest of the acting phase: normaliser updates, value pass, GAE
norm_p = ppo.RunningNorm(abi.PRIV, dev)
cfg = ppo.PPOConfig(); T, n = 40, 4096
batch = a.S
def timed(label, fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print(f"  {label:28s} {(time.perf_counter() - t0) / reps * 1e3:7.2f} ms"); return r
with torch.no_grad():
    timed("norm_s.update", lambda: ns.update(batch["obs"]))
    timed("norm_p.update", lambda: norm_p.update(batch["priv"]))
    last_priv = env.buffers["obs_priv"].clone()
    values = timed("value pass", lambda: model.value(norm_p(torch.cat([batch["priv"], last_priv[None]], 0))).squeeze(-1))
    def gae():
        term = batch["done"] * (1.0 - batch["trunc"])
        adv = torch.zeros_like(batch["rew"]); last = torch.zeros(n, device=dev)
        for t in reversed(range(T)):
            nonterm = 1.0 - term[t]
            delta = batch["rew"][t] + cfg.discounting * values[t + 1] * nonterm - values[t]
            last = delta + cfg.discounting * cfg.gae_lambda * nonterm * (1.0 - batch["done"][t]) * last
            adv[t] = last
        return adv
    timed("GAE loop", gae)
