"""Inference player (``test=True`` of train.py:166-171; deployment consumer
isaacgymenvs/vine_robot_test_model.py:143-177): loads a checkpoint and runs the deterministic policy."""
import torch

from .network import ModelA2CContinuousLogStd


class PpoPlayerContinuous:
    def __init__(self, params, vec_env=None):
        self.config = config = params["config"]
        self.vec_env = vec_env
        from ..utils.rlgames_utils import RLGPUEnv
        if self.vec_env is None:
            self.vec_env = RLGPUEnv(config["env_name"], config["num_actors"])
        elif not hasattr(self.vec_env, "get_env_info"):
            self.vec_env = RLGPUEnv.wrap(self.vec_env)
        info = self.vec_env.get_env_info()
        self.device = torch.device(config.get("device", "cuda:0"))
        self.actions_num = info["action_space"].shape[0]
        self.model = ModelA2CContinuousLogStd(params["network"], self.actions_num, tuple(info["observation_space"].shape),
                                              config.get("normalize_value", False), config["normalize_input"]).to(self.device)
        self.model.eval()
        self.is_deterministic = config.get("player", {}).get("deterministic", True)
        self.max_steps = config.get("player", {}).get("max_steps", 27000)
        self.states = None

    def restore(self, fn):
        ckpt = torch.load(fn, map_location=self.device, weights_only=False)
        self.model.load_state_dict(ckpt["model"])

    def init_rnn(self, batch):
        self.states = [s.clone() for s in self.model.get_default_rnn_state(batch, self.device)]

    @torch.no_grad()
    def get_action(self, obs, is_deterministic=True):
        if self.states is None:
            self.init_rnn(obs.shape[0])
        res = self.model({"is_train": False, "prev_actions": None, "obs": obs, "rnn_states": self.states})
        self.states = list(res["rnn_states"])
        action = res["mus"] if is_deterministic else res["actions"]
        return torch.clamp(action, -1.0, 1.0)

    def run(self, n_steps=None):
        """Plays ``n_steps`` env steps (default: max_steps); returns mean episode return and length."""
        n_steps = n_steps or self.max_steps
        obs = self.vec_env.reset()["obs"].to(self.device)
        n = obs.shape[0]
        cur_r = torch.zeros(n, device=self.device)
        cur_l = torch.zeros(n, device=self.device)
        sum_r = torch.zeros((), device=self.device)
        sum_l = torch.zeros((), device=self.device)
        games = torch.zeros((), device=self.device)
        for _ in range(n_steps):
            action = self.get_action(obs, self.is_deterministic)
            o, r, d, _ = self.vec_env.step(action)
            obs = o["obs"].to(self.device)
            d = d.to(self.device).float()
            cur_r += r.to(self.device)
            cur_l += 1
            sum_r += (cur_r * d).sum()
            sum_l += (cur_l * d).sum()
            games += d.sum()
            keep = 1.0 - d
            self.states = [s * keep.view(1, -1, 1) for s in self.states]
            cur_r *= keep
            cur_l *= keep
        g = max(float(games), 1.0)
        print("reward:", float(sum_r) / g, "steps:", float(sum_l) / g, "games:", int(games))
        return float(sum_r) / g, float(sum_l) / g
