"""the pieces of the PPO update that touch the env / policy contract
(IsaacGymEnvs/algorithms/ppo_asymmetry.py:142-175 schedules, :398-404 spectral-norm clamp, :458-468 TorchScript actor export).

These run once per epoch on the host (scalars) or on the caller's torch module (training side); nothing here is on the step path.
`Schedules.at(epoch)` returns the three values the reference computes at the top of `update()`; the caller writes them where the
reference does: `optimizer.param_groups[0]['lr']`, `spectral_normalize_actor_(agent.actor_mlp, lipschitz)`, `env.difficulty = d`
(which reaches the HIP kernel through taco_set_difficulty).
"""
import torch
import torch.nn as nn


def _ramp(x, x0, x1, y0, y1):
    """y0 before x0, y1 after x1, the straight line in between (the reference's three-way conditional, :154-160 / :166-171)"""
    if x < x0:
        return y0
    if x > x1:
        return y1
    return (y1 - y0) / (x1 - x0) * (x - x0) + y0


class Schedules:
    """Defaults = the reference's constructor defaults (ppo_asymmetry.py:28-32, 78-97)."""

    def __init__(self, epochs=500, lr=3e-4, learning_rate_schedule=True, lr_ratio=0.3, lr_lp_index=0.7, lr_epoch_index=350,
                 lipschitz_para=5, lipschitz_schedule=True, lip_ratio=None, lip_lp_index=None, lip_epoch_index=None,
                 difficulty_schedule=True, diff_value=None, diff_lp_index=None, diff_epoch_index=None):
        self.epochs, self.lr = epochs, lr
        self.learning_rate_schedule, self.lr_ratio, self.lr_lp_index, self.lr_epoch_index = learning_rate_schedule, lr_ratio, lr_lp_index, lr_epoch_index
        self.lipschitz_para, self.lipschitz_schedule = lipschitz_para, lipschitz_schedule
        self.lip_ratio = [1, 0.3] if lip_ratio is None else lip_ratio
        self.lip_lp_index = [0.3, 0.7] if lip_lp_index is None else lip_lp_index
        self.lip_epoch_index = [100, 500] if lip_epoch_index is None else lip_epoch_index
        self.difficulty_schedule = difficulty_schedule
        self.diff_value = [0.1, 1] if diff_value is None else diff_value
        self.diff_lp_index = [0.3, 0.7] if diff_lp_index is None else diff_lp_index
        self.diff_epoch_index = [100, 500] if diff_epoch_index is None else diff_epoch_index

    def learning_rate(self, epoch):
        """:142-150: linear decay to lr_ratio, whichever of the two clocks (learning process, epoch) is further along"""
        lp = epoch / self.epochs
        if not self.learning_rate_schedule:
            return self.lr_ratio * self.lr
        r0 = (self.lr_ratio - 1) / self.lr_lp_index * lp + 1 if lp < self.lr_lp_index else self.lr_ratio
        r1 = (self.lr_ratio - 1) / self.lr_epoch_index * epoch + 1 if epoch < self.lr_epoch_index else self.lr_ratio
        return min(r0, r1) * self.lr

    def lipschitz(self, epoch):
        """:152-162"""
        lp = epoch / self.epochs
        if not self.lipschitz_schedule:
            return self.lip_ratio[1] * self.lipschitz_para
        p0 = _ramp(lp, self.lip_lp_index[0], self.lip_lp_index[1], self.lip_ratio[0], self.lip_ratio[1])
        p1 = _ramp(epoch, self.lip_epoch_index[0], self.lip_epoch_index[1], self.lip_ratio[0], self.lip_ratio[1])
        return min(p0, p1) * self.lipschitz_para

    def difficulty(self, epoch):
        """:164-175"""
        lp = epoch / self.epochs
        if not self.difficulty_schedule:
            return self.diff_value[1]
        d0 = _ramp(lp, self.diff_lp_index[0], self.diff_lp_index[1], self.diff_value[0], self.diff_value[1])
        d1 = _ramp(epoch, self.diff_epoch_index[0], self.diff_epoch_index[1], self.diff_value[0], self.diff_value[1])
        return max(d0, d1)

    def at(self, epoch):
        return self.learning_rate(epoch), self.lipschitz(epoch), self.difficulty(epoch)


@torch.no_grad()
def spectral_normalize_actor_(actor_mlp, lipschitz_const=5):
    """:398-404: every weight matrix of the actor MLP whose largest singular value exceeds the constant is scaled back onto it.
    `actor_mlp` is an nn.Module (its parameters are scaled in place) or a state_dict-like mapping of tensors."""
    params = actor_mlp.parameters() if isinstance(actor_mlp, nn.Module) else [v for k, v in actor_mlp.items() if k.startswith("actor_mlp.") or "." not in k.split("layers")[0]]
    n = 0
    for p in params:
        if p.ndim > 1:
            s = torch.linalg.matrix_norm(p, ord=2)
            if s > lipschitz_const:
                p.mul_(lipschitz_const / s)
                n += 1
    return n


def actor_module_from_state_dict(state_dict, len_obs, num_obs=26):
    """The actor as a plain torch module with the reference's forward semantics (nets_asymmetry.py:37-39, :380-387):
    x.view(N, -1) -> Linear / ReLU ... -> Linear -> Tanh."""
    idx = sorted({int(k.split(".")[2]) for k in state_dict if k.startswith("actor_mlp.layers.") and k.endswith(".weight")})
    layers = []
    for j, i in enumerate(idx):
        w = torch.as_tensor(state_dict[f"actor_mlp.layers.{i}.weight"]).float().cpu()
        b = torch.as_tensor(state_dict[f"actor_mlp.layers.{i}.bias"]).float().cpu()
        lin = nn.Linear(w.shape[1], w.shape[0])
        with torch.no_grad():
            lin.weight.copy_(w)
            lin.bias.copy_(b)
        layers += [lin, nn.Tanh() if j == len(idx) - 1 else nn.ReLU()]

    class Actor(nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = nn.Sequential(*layers)

        def forward(self, x):
            return self.layers(x.contiguous().view(x.size(0), -1))

    m = Actor().eval()
    assert m.layers[0].in_features == len_obs * num_obs, "state_dict does not match len_obs x num_obs"
    return m


def save_actor_as_pt(state_dict, path, len_obs, num_obs=26, device="cpu"):
    """:458-468: trace the actor on a zero observation and save the TorchScript module (the on-board deployment format).
    Returns (traced module, its output on the zero observation)."""
    actor = actor_module_from_state_dict(state_dict, len_obs, num_obs).to(device)
    obs = torch.zeros((1, len_obs, num_obs), device=device)
    traced = torch.jit.trace(actor, obs)
    traced.save(path)
    return traced, traced(obs)
