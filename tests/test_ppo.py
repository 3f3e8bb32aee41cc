"""PPO arithmetic (rows R1-R6): rl-games==1.5.2 is absent (parity unpinned) -- these tests pin each
formula this build chose against a direct restatement, and the host-side mechanics on CPU."""
import math
import os

import numpy as np
import pytest
import torch

from vine_robot_isaacgymenvs_amd import load_config
from vine_robot_isaacgymenvs_amd.learning import a2c_continuous as a2c
from vine_robot_isaacgymenvs_amd.learning.network import ModelA2CContinuousLogStd
from vine_robot_isaacgymenvs_amd.learning.running_mean_std import RunningMeanStd


def make_agent(num_envs=32, minibatch=128, seed=42, **conf):
    from oracle.oracle_vec_task import OracleVecTask
    cfg = load_config(overrides=["num_envs=%d" % num_envs, "minibatch_size=%d" % minibatch, "rl_device=cpu"])
    env = OracleVecTask(cfg["task"], seed=seed)
    params = cfg["train"]["params"]
    params["config"].update(write_files=False, print_stats=False, **conf)
    torch.manual_seed(seed)
    return a2c.A2CAgent("t", params, vec_env=env), cfg


def test_network_matches_config_and_param_count():
    cfg = load_config()
    for nobs, expect in ((28, 408261), (18, 395461)):
        m = ModelA2CContinuousLogStd(cfg["train"]["params"]["network"], 2, (nobs,), True, True)
        assert sum(p.numel() for p in m.parameters()) == expect
    keys = set(m.state_dict())
    for k in ("a2c_network.sigma", "a2c_network.actor_mlp.0.weight", "a2c_network.rnn.rnn.weight_ih_l0",
              "a2c_network.layer_norm.weight", "a2c_network.mu.weight", "a2c_network.value.bias",
              "running_mean_std.running_mean", "value_mean_std.count"):
        assert k in keys
    assert float(m.a2c_network.sigma.abs().max()) == 0.0
    assert all(float(l.bias.abs().max()) == 0.0 for l in m.a2c_network.modules() if isinstance(l, torch.nn.Linear))


def test_lstm_step_loop_equals_nn_lstm_and_zeroes_on_done():
    cfg = load_config()
    m = ModelA2CContinuousLogStd(cfg["train"]["params"]["network"], 2, (28,), True, True)
    w = m.a2c_network.rnn
    ref = torch.nn.LSTM(92, 256, 1)
    ref.load_state_dict(w.rnn.state_dict())
    x = torch.randn(4, 6, 92)
    h0 = (torch.randn(1, 6, 256), torch.randn(1, 6, 256))
    o1, (h1, c1) = ref(x, h0)
    o2, (h2, c2) = w(x, h0)
    assert torch.allclose(o1, o2, atol=1e-6) and torch.allclose(c1, c2, atol=1e-6)
    dones = torch.zeros(4, 6)
    dones[2, 3] = 1                       # env 3 finished an episode before step 2
    o3, _ = w(x, h0, dones)
    assert torch.allclose(o3[:2], o2[:2]) and torch.allclose(o3[:, :3], o2[:, :3])
    z = (torch.zeros(1, 1, 256), torch.zeros(1, 1, 256))
    o4, _ = ref(x[2:, 3:4], z)
    assert torch.allclose(o3[2:, 3:4], o4, atol=1e-6)


def test_running_mean_std_matches_numpy():
    rms = RunningMeanStd((5,))
    rms.train()
    rng = np.random.default_rng(0)
    chunks = [rng.normal(2.0, 3.0, (n, 5)).astype(np.float32) for n in (7, 100, 33)]
    for c in chunks:
        y = rms(torch.from_numpy(c))
    allx = np.concatenate([np.zeros((1, 5), np.float32)] + chunks)    # count starts at 1 with mean 0 / var 1
    cnt = sum(len(c) for c in chunks) + 1
    assert float(rms.count) == cnt
    # reproduce the merge sequence in float64
    mean, var, count = np.zeros(5), np.ones(5), 1.0
    for c in chunks:
        bm, bv, bc = c.mean(0), c.var(0, ddof=1), len(c)
        delta = bm - mean
        tot = count + bc
        m2 = var * count + bv * bc + delta ** 2 * count * bc / tot
        mean, var, count = mean + delta * bc / tot, m2 / tot, tot
    np.testing.assert_allclose(rms.running_mean.numpy(), mean, rtol=1e-6)
    np.testing.assert_allclose(rms.running_var.numpy(), var, rtol=1e-6)
    rms.eval()
    x = torch.from_numpy(chunks[0])
    y = rms(x)
    np.testing.assert_allclose(y.numpy(), np.clip((chunks[0] - mean) / np.sqrt(var + 1e-5), -5, 5), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rms(y, unnorm=True).numpy(), chunks[0], rtol=1e-4, atol=1e-4)
    assert float(rms.count) == cnt       # eval mode does not update


def test_gae_next_nonterminal_form():
    T, N = 6, 5
    g = torch.Generator().manual_seed(0)
    rew, val = torch.rand(T, N, 1, generator=g), torch.rand(T, N, 1, generator=g)
    dones = (torch.rand(T, N, generator=g) < 0.3).float()
    fd, last = (torch.rand(N, generator=g) < 0.3).float(), torch.rand(N, 1, generator=g)
    adv = a2c.discount_values(0.99, 0.95, fd, last, dones, val, rew)
    exp = np.zeros((T, N))
    for e in range(N):
        lam = 0.0
        for t in reversed(range(T)):
            nnt = 1 - (fd[e] if t == T - 1 else dones[t + 1, e]).item()
            nv = (last[e, 0] if t == T - 1 else val[t + 1, e, 0]).item()
            delta = rew[t, e, 0].item() + 0.99 * nv * nnt - val[t, e, 0].item()
            lam = delta + 0.99 * 0.95 * nnt * lam
            exp[t, e] = lam
    np.testing.assert_allclose(adv[..., 0].numpy(), exp, rtol=1e-5, atol=1e-6)


def test_losses_and_kl():
    adv = torch.tensor([1.0, -1.0, 2.0, -0.5])
    old, new = torch.tensor([1.0, 1.0, 1.0, 1.0]), torch.tensor([0.5, 1.6, 1.0, 0.9])
    ratio = torch.exp(old - new)
    exp = torch.max(-adv * ratio, -adv * ratio.clamp(0.8, 1.2))
    assert torch.allclose(a2c.actor_loss(old, new, adv, 0.2), exp)
    vp, v, r = torch.tensor([[0.0], [1.0]]), torch.tensor([[0.5], [0.9]]), torch.tensor([[1.0], [0.0]])
    cl = a2c.critic_loss(vp, v, 0.2, r, True)
    assert torch.allclose(cl, torch.tensor([[max(0.25, 0.64)], [max(0.81, 0.81)]]))
    assert torch.allclose(a2c.critic_loss(vp, v, 0.2, r, False), (r - v) ** 2)
    mu = torch.tensor([[1.3, -1.0], [0.0, -1.5]])
    assert torch.allclose(a2c.bound_loss(mu), torch.tensor([0.2 ** 2, 0.4 ** 2]), atol=1e-6)
    m0, s0 = torch.tensor([[0.1, -0.2]]), torch.tensor([[1.0, 0.5]])
    m1, s1 = torch.tensor([[0.0, 0.1]]), torch.tensor([[0.8, 0.7]])
    exact = (torch.log(s1 / s0) + (s0 ** 2 + (m1 - m0) ** 2) / (2 * s1 ** 2) - 0.5).sum()
    assert abs(float(a2c.policy_kl(m0, s0, m1, s1)) - float(exact)) < 1e-4      # +1e-5 guards only
    assert abs(float(a2c.policy_kl(m0, s0, m0, s0))) < 2e-5


def test_loss_terms_match_reference_text(golden):
    """Golden F8: actor / critic / bound loss per sample as the reference's in-tree restatement computes them
    (isaacgymenvs/learning/common_agent.py:482-516, 427-435; its bound uses 1.0 where rl_games 1.5.2 uses 1.1)."""
    g = golden("f8_ppo_loss_terms")
    t = lambda k: torch.from_numpy(g[k])
    e = float(g["e_clip"])
    np.testing.assert_allclose(a2c.actor_loss(t("old_neglogp"), t("neglogp"), t("advantage"), e).numpy(), g["a_loss"],
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(a2c.critic_loss(t("old_values"), t("values"), e, t("returns"), True).numpy(), g["c_loss"],
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(a2c.critic_loss(t("old_values"), t("values"), e, t("returns"), False).numpy(),
                               g["c_loss_noclip"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(a2c.bound_loss(t("mu"), soft_bound=1.0).numpy(), g["b_loss_soft_bound_1"], rtol=1e-6,
                               atol=1e-7)
    assert (g["b_loss_soft_bound_1"] > 0).mean() > 0.1 and (np.abs(g["values"] - g["old_values"]) > e).any()


def test_neglogp_is_gaussian_nll():
    mu, logstd = torch.tensor([[0.3, -0.1]]), torch.tensor([[0.2, -0.4]])
    x = torch.tensor([[0.5, 0.5]])
    d = torch.distributions.Normal(mu, logstd.exp())
    assert torch.allclose(ModelA2CContinuousLogStd.neglogp(x, mu, logstd.exp(), logstd), -d.log_prob(x).sum(-1), atol=1e-6)


def test_adaptive_lr_legacy_schedule():
    agent, _ = make_agent()
    lr0 = float(agent.lr)
    agent.update_lr_from_kl(torch.tensor(0.02))         # > 2 * 0.008
    assert float(agent.lr) == pytest.approx(lr0 / 1.5)
    agent.update_lr_from_kl(torch.tensor(0.001))        # < 0.5 * 0.008
    assert float(agent.lr) == pytest.approx(lr0)
    agent.update_lr_from_kl(torch.tensor(0.008))
    assert float(agent.lr) == pytest.approx(lr0)
    agent.lr.fill_(1e-6); agent.update_lr_from_kl(torch.tensor(1.0))
    assert float(agent.lr) == pytest.approx(1e-6)       # floor
    agent.lr.fill_(1e-2); agent.update_lr_from_kl(torch.tensor(0.0))
    assert float(agent.lr) == pytest.approx(1e-2)       # ceiling
    assert agent.optimizer.param_groups[0]["lr"] is agent.lr     # Adam reads the device scalar


def test_rollout_buffers_and_dataset_layout():
    agent, _ = make_agent(num_envs=8, minibatch=32)
    agent.init_tensors()
    agent.obs = agent.env_reset()["obs"]
    agent.set_eval()
    with torch.no_grad():
        batch = agent.play_steps_rnn()
    T, N = agent.horizon_length, agent.num_actors
    assert batch["obses"].shape == (T * N, 28) and batch["returns"].shape == (T * N, 1)
    # env-major flattening: row env*T + t
    assert torch.equal(batch["obses"][3 * T + 5], agent.buf["obses"][5, 3])
    assert batch["rnn_states"][0].shape == (1, N * T // agent.seq_len, 256)
    # stored LSTM state of chunk c of env e sits at sequence index e*(T/seq)+c
    # (the rollout stores them as [layer, env, chunk, H], i.e. in dataset order already: the dataset tensor is a view)
    assert torch.equal(batch["rnn_states"][1][0, 3 * (T // 4) + 2], agent.mb_rnn_states[1][0, 3, 2])
    assert batch["rnn_states"][1].data_ptr() == agent.mb_rnn_states[1].data_ptr()
    # first stored dones are the initial ones (rl_games starts with dones = 1)
    assert bool((agent.buf["dones"][0] == 1).all())
    # reward shaper (scale 0.01) applied; bootstrap only where time_outs
    agent.set_train()
    agent.prepare_dataset(dict(batch))
    mb = agent.get_minibatch(1)
    assert mb["range"] == (32, 64) and mb["obs"].shape == (32, 28) and mb["rnn_states"][0].shape == (1, 8, 256)
    adv = agent.dataset["advantages"]
    assert abs(float(adv.mean())) < 1e-5 and abs(float(adv.std()) - 1.0) < 1e-4


def test_flat_gradient_buffer_views():
    agent, _ = make_agent(num_envs=8, minibatch=32)
    assert agent.num_params == 408261 and agent.flat_grads.numel() >= 408261
    for p, off in zip(agent.optimizer.params, agent.optimizer.offsets):
        assert p.grad.data_ptr() == agent.flat_grads.data_ptr() + 4 * off and off % 64 == 0
        assert p.data_ptr() == agent.optimizer.flat_params.data_ptr() + 4 * off


@pytest.mark.parametrize("scale", [None, 65536.0])
@pytest.mark.parametrize("gain", [0.01, 30.0])
def test_clip_grad_norm_under_loss_scaling(scale, gain):
    """``truncate_grads: True`` with the device-side GradScaler: the gradient block holds scale * g until the
    Adam kernel unscales it, and the clip threshold applies to the UNSCALED norm -- what rl_games gets from
    ``scaler.unscale_`` followed by ``clip_grad_norm_``."""
    from vine_robot_isaacgymenvs_amd.learning.flat_adam import FlatAdam
    torch.manual_seed(3)
    params = [torch.nn.Parameter(torch.randn(33, 7)), torch.nn.Parameter(torch.randn(130))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt = FlatAdam(params, lr=1e-3)
    if scale is not None:
        opt.enable_loss_scaling(init_scale=scale)
    s = 1.0 if scale is None else scale
    for p, r in zip(params, ref):
        g = torch.randn_like(r) * gain
        r.grad = g.clone()
        p.grad.copy_(g * s)                               # what the loss kernels leave behind
    expect = torch.nn.utils.clip_grad_norm_(ref, 1.0)     # (the unscaled norm)
    got = opt.clip_grad_norm_(1.0)
    assert abs(float(got) - float(expect)) <= 1e-5 * float(expect)
    assert (float(expect) > 1.0) == (gain > 1.0)          # one case clips, the other does not
    for p, r in zip(params, ref):
        torch.testing.assert_close(p.grad / s, r.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_clip_grad_norm_with_a_non_finite_gradient_flags_the_overflow(bad):
    """an inf / NaN gradient gives a non-finite norm and coef = 0 / NaN -- without the flag Adam would take a
    zero-gradient step (moments decay) where rl_games' unscale_ + clip + scaler.step skips.  The flag must be raised by the
    clip itself, whoever delivered the gradients (check_grads off, bypassed batch)."""
    from vine_robot_isaacgymenvs_amd.learning.flat_adam import FlatAdam
    torch.manual_seed(5)
    params = [torch.nn.Parameter(torch.randn(40, 3)), torch.nn.Parameter(torch.randn(17))]
    opt = FlatAdam(params, lr=1e-3)
    opt.enable_loss_scaling(init_scale=1024.0)
    opt.check_grads = False
    for p in params:
        p.grad.copy_(torch.randn_like(p) * 1024.0)
    assert float(opt.found_inf) == 0.0
    opt.clip_grad_norm_(1.0)
    assert float(opt.found_inf) == 0.0                      # finite gradients: the flag stays down
    params[0].grad[3, 1] = bad
    norm = opt.clip_grad_norm_(1.0)
    assert not torch.isfinite(norm) and float(opt.found_inf) >= 1.0


def test_training_runs_and_checkpoint_roundtrip(tmp_path):
    agent, cfg = make_agent(num_envs=16, minibatch=64, max_epochs=2)
    agent.nn_dir = str(tmp_path)
    before = [p.detach().clone() for p in agent.model.parameters()]
    agent.train()
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.model.parameters()))
    assert all(torch.isfinite(p).all() for p in agent.model.parameters())
    assert float(agent.model.running_mean_std.count) > 1 and float(agent.model.value_mean_std.count) > 1
    path = agent.save(os.path.join(str(tmp_path), "ck"))
    agent2, _ = make_agent(num_envs=16, minibatch=64, seed=7)
    agent2.restore(path)
    for a, b in zip(agent.model.state_dict().values(), agent2.model.state_dict().values()):
        assert torch.equal(a, b)
    assert agent2.epoch_num == 2 and float(agent2.lr) == pytest.approx(float(agent.lr))
    # the player consumes the same checkpoint (vine_robot_test_model.py's use)
    from vine_robot_isaacgymenvs_amd.learning.player import PpoPlayerContinuous
    params = cfg["train"]["params"]
    player = PpoPlayerContinuous(params, vec_env=agent.vec_env)
    player.restore(path)
    act = player.get_action(torch.zeros(16, 28))
    assert act.shape == (16, 2) and float(act.abs().max()) <= 1.0
    # deployment loader (vine_robot_test_model.py:143-177): config pickle + .pth -> obs -> action, no simulator
    import pickle
    from vine_robot_isaacgymenvs_amd.vine_robot_test_model import VinePolicy
    pkl = os.path.join(str(tmp_path), "cfg.pkl")
    with open(pkl, "wb") as f:
        pickle.dump(cfg["train"], f)
    pol = VinePolicy.load(pkl, path)
    a1 = pol.get_action(np.zeros(28, np.float32))
    assert a1.shape == (2,) and torch.allclose(a1, act[0], atol=1e-6)
    a2 = pol.get_action(np.zeros(28, np.float32))       # the LSTM state carries over between calls
    pol.reset()
    assert torch.allclose(pol.get_action(np.zeros(28, np.float32)), a1, atol=1e-6) and not torch.equal(a1, a2)


def test_baseline_config1_cpu_reference_path():
    """BASELINE.json configs[0]: 64 envs on the CPU path, 1 PPO iteration (minibatch 1024 = 64 x 16, since the
    default 32768 does not divide it -- SURVEY 8d).  Runs the oracle-backed env + the same agent code."""
    agent, _ = make_agent(num_envs=64, minibatch=1024, max_epochs=1)
    assert agent.num_minibatches == 1 and agent.batch_size == 1024
    last, epoch = agent.train()
    assert epoch == 1 and agent.frame == 1024
    assert all(torch.isfinite(p).all() for p in agent.model.parameters())


def test_config_constraints_are_asserted():
    with pytest.raises(AssertionError, match="minibatch_size"):
        make_agent(num_envs=64, minibatch=32768)       # 64*16 = 1024 is not a multiple of 32768 (BASELINE config 1 note)


@pytest.mark.parametrize("num_actions", [2, 1])
def test_robot_side_control_model(tmp_path, num_actions):
    """N2: the reference's deployment entry point (isaacgymenvs/vine_robot_test_model.py:143-177) -- constructor
    (config pickle, checkpoint, x_range, u_range), get_action(q, qd, tip_pos, tip_vel, target_pos) on 1-D tensors
    (N_OBS = 19 there), [-1, 1] -> (rail range, u range) by (x + 1) (high - low) / 2 + low, LSTM state carried
    between calls -- on a checkpoint + config pickle written the way train.py writes them."""
    import pickle
    from vine_robot_isaacgymenvs_amd.vine_robot_test_model import VineRobotControlModel
    cfg = load_config()
    params = cfg["train"]["params"]
    torch.manual_seed(3)
    net = ModelA2CContinuousLogStd(params["network"], num_actions, (19,), True, True)
    with torch.no_grad():
        net.a2c_network.mu.weight.normal_(0, 0.03)         # make the mean depend on the observation visibly
        net.running_mean_std.running_mean.normal_(0, 0.1)
    torch.save({"model": net.state_dict(), "epoch": 7}, tmp_path / "p.pth")
    with open(tmp_path / "cfg.pkl", "wb") as f:
        pickle.dump({"params": params}, f)
    x_range, u_range = (-10.0, 10.0), (-0.1, 3.0)
    m = VineRobotControlModel(str(tmp_path / "cfg.pkl"), str(tmp_path / "p.pth"), x_range, u_range, deterministic=True).to("cpu")
    g = torch.Generator().manual_seed(0)
    args = [torch.randn(5, generator=g) * 0.2, torch.randn(5, generator=g), torch.randn(3, generator=g) * 0.3,
            torch.randn(3, generator=g), torch.tensor([0.0, -0.4, 0.6])]
    a1 = m.get_action(*args)
    a2 = m.get_action(*args)                                # the LSTM state moved on: same input, different output
    assert a1.shape == (num_actions,) and not torch.equal(a1, a2)
    # against the network driven by hand
    net.eval()
    st = [s.clone() for s in net.get_default_rnn_state(1, "cpu")]
    with torch.no_grad():
        mu = torch.clamp(net({"is_train": False, "prev_actions": None, "obs": torch.cat(args)[None], "rnn_states": st})["mus"][0], -1, 1)
    if num_actions == 2:
        want = torch.stack([(mu[0] + 1) * 20.0 / 2 - 10.0, (mu[1] + 1) * 3.1 / 2 - 0.1])
        assert x_range[0] <= float(a1[0]) <= x_range[1]
    else:
        want = (mu + 1) * 3.1 / 2 - 0.1
    assert torch.allclose(a1, want, atol=1e-6) and u_range[0] <= float(a1[-1]) <= u_range[1]
    assert float(m.rescale(torch.tensor(-1.0), 2.0, 6.0)) == 2.0 and float(m.rescale(torch.tensor(1.0), 2.0, 6.0)) == 6.0
    m.reset()
    assert torch.equal(m.get_action(*args), a1)             # episode start: state forgotten
    # sampled actions (the reference player's default) stay inside the ranges
    s = VineRobotControlModel(str(tmp_path / "cfg.pkl"), str(tmp_path / "p.pth"), x_range, u_range)
    for _ in range(20):
        a = s.get_action(*args)
        assert u_range[0] - 1e-5 <= float(a[-1]) <= u_range[1] + 1e-5
    with pytest.raises(ValueError):
        m.get_action(args[0], args[1], args[2], args[3], torch.zeros(5))


def test_gae_against_reference_text_golden(golden):
    """F9: the reference's in-tree ``discount_values`` (common_agent.py:413-425, lifted by name into the fixture
    generator) on T + 1 terminal flags / values; this build's next-nonterminal form (rl_games') must give the same
    advantages -- pins row R2 against the only GAE text the reference holds."""
    import torch
    from vine_robot_isaacgymenvs_amd.learning import a2c_continuous as a2c
    g = golden("f9_gae")
    T = g["rewards"].shape[0]
    t = lambda k: torch.from_numpy(g[k])
    dones, values = t("dones"), t("values")
    adv = a2c.discount_values(float(g["gamma"]), float(g["tau"]), dones[T], values[T], dones[:T], values[:T], t("rewards"))
    assert float((adv - t("advs")).abs().max()) < 2e-6
    assert float(dones.sum()) > 50 and float(t("advs").abs().max()) > 1.0


def test_rollout_bookkeeping_against_reference_text_golden(golden):
    """F10: the reference's in-tree ``play_steps`` (common_agent.py:257-316, lifted by name into the fixture generator and
    run on replayed rewards / done flags): the episode accumulators after every step and the finished episodes handed to
    the meters.  The product's ``A2CAgent._episode_bookkeeping`` (the stock rollout's five lines) must keep the same
    books; its masked meter must have seen exactly the reference's finished episodes (sum and count per step)."""
    import types
    import torch
    from vine_robot_isaacgymenvs_amd.learning import a2c_continuous as a2c
    g = golden("f10_rollout_bookkeeping")
    T, N = g["dones"].shape

    class Recorder(a2c.DeviceAverageMeter):
        def __init__(self):
            super().__init__(1, 100, "cpu")
            self.seen = []

        def update(self, values, mask):
            self.seen.append((float((values.double() * mask.double().unsqueeze(-1)).sum()), int(mask.sum())))
            super().update(values, mask)

    agent = types.SimpleNamespace(current_rewards=torch.zeros(N, 1), current_lengths=torch.zeros(N),
                                  game_rewards=Recorder(), game_lengths=Recorder())
    for t in range(T):
        rew, done = torch.from_numpy(g["rewards"][t]), torch.from_numpy(g["dones"][t])
        nd = a2c.A2CAgent._episode_bookkeeping(agent, rew, done)
        assert torch.equal(nd, 1.0 - done.float())
        assert float((agent.current_rewards - torch.from_numpy(g["cur_rewards"][t])).abs().max()) < 1e-5
        assert torch.equal(agent.current_lengths, torch.from_numpy(g["cur_lengths"][t]))
        assert agent.game_rewards.seen[t][1] == int(g["finished_count"][t]) == agent.game_lengths.seen[t][1]
        assert abs(agent.game_rewards.seen[t][0] - float(g["finished_return_sum"][t])) < 1e-3
        assert agent.game_lengths.seen[t][0] == float(g["finished_length_sum"][t])
    # the shaper of the fixture is the YAML's scale_value (PY:28) on the raw reward; done flags go to the buffer as they are
    assert np.allclose(g["shaped"], g["rewards"] * float(g["reward_scale"]), atol=1e-7)
    assert np.array_equal(g["buffer_dones"], g["dones"].astype(np.float32))
    assert int(g["finished_count"][3]) >= 34 and int(g["finished_count"][7]) == 0
