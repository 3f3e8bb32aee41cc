"""Multi-GPU path rehearsed on CPU: 2 processes, gloo, one env shard per rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, same_data, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from tests.test_ppo import make_agent
    seed = 42 if same_data else 42 + 2 * rank            # train.py:78 + utils.py:50 -> 42 + 2*rank
    agent, _ = make_agent(num_envs=16, minibatch=64, seed=seed, multi_gpu=world > 1, max_epochs=2)
    if not same_data:
        torch.manual_seed(1000 + rank)                   # different initial weights: the broadcast must fix that
        for p in agent.model.parameters():
            p.data.add_(0.01 * torch.randn_like(p))
    agent.train()
    flat = torch.cat([p.detach().flatten() for p in agent.model.parameters()])
    out[rank] = (flat, float(agent.lr), agent.frame)
    if world > 1:
        dist.destroy_process_group()


def _run(world, same_data):
    mgr = mp.get_context("spawn").Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, same_data, out), nprocs=world, join=True)
    return dict(out)


@pytest.mark.timeout(600)
def test_two_ranks_stay_in_lockstep():
    """Different env shards and different initial weights per rank: after the rank-0 broadcast and two
    iterations of all-reduced gradients + all-reduced KL the replicas are bit-identical."""
    out = _run(2, same_data=False)
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]                         # adaptive LR decided from the averaged KL
    assert out[0][2] == 2 * 2 * 16 * 16                   # frames count both shards


@pytest.mark.timeout(600)
def test_gradient_average_of_identical_shards_equals_single_process():
    """SUM / world of identical gradients == the single-process gradient: 2 ranks fed the same data reproduce
    the 1-rank run (up to the reduction's rounding)."""
    two = _run(2, same_data=True)
    one = _run(1, same_data=True)
    assert torch.allclose(two[0][0], one[0][0], rtol=1e-5, atol=1e-6)
    assert two[0][1] == pytest.approx(one[0][1])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_multi_gpu_code_path_on_one_rank(monkeypatch):
    """The code every rank of a multi-GPU job executes, on the one GPU a test box has: RCCL world of ONE rank.
    ``multi_gpu=True`` makes the agent pin its device, ``init_process_group("nccl", device_id=...)``, broadcast the
    parameters, and run every optimiser step as two hipGraph replays around the EAGER ``all_reduce(comm_buffer)`` with
    the collective capture decision in front (a 1-rank run without ``multi_gpu`` replays whole mini-epochs instead).
    The result must equal the plain single-rank run: SUM over one rank / 1."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from vine_robot_isaacgymenvs_amd import load_config
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import A2CAgent
    from vine_robot_isaacgymenvs_amd.tasks import isaacgym_task_map
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1").items():
        monkeypatch.setenv(k, v)
    results = []
    try:
        for multi, in_graph in ((True, True), (True, False), (False, True)):
            cfg = load_config(overrides=["num_envs=256", "minibatch_size=1024"])
            cfg["task"]["seed"] = 42
            env = isaacgym_task_map["Vine5LinkMovingBase"](cfg=cfg["task"], rl_device="cuda:0", sim_device="cuda:0",
                                                          graphics_device_id=0, headless=True)
            params = cfg["train"]["params"]
            params["config"].update(write_files=False, print_stats=False, use_graphs=True, multi_gpu=multi, device="cuda",
                                    collective_in_graph=in_graph)
            torch.manual_seed(0)
            agent = A2CAgent("t", params, vec_env=env)
            assert torch.cuda.current_device() == agent.device.index == 0
            assert agent.multi_gpu == multi and (dist.is_initialized() or not multi)
            if multi:
                assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
            agent.init_tensors()
            agent.obs = agent.env_reset()["obs"]
            agent.broadcast_parameters()
            torch.manual_seed(123)
            for _ in range(3):
                agent.train_epoch()
            torch.cuda.synchronize()
            assert agent.graph_status["rollout"] == "graph", agent.graph_status
            if multi:
                # round 4: the collective is captured into the mini-epoch graph when RCCL accepts a capture on this stack
                # (probed once); refused -> one graph per optimiser step with the eager all-reduce between them
                cc = agent.collective_capture
                print("collective capture:", cc, agent.graph_status["update"])
                assert agent.graph_status["update"].startswith(
                    "graph (1 per iteration, all-reduces captured" if cc["in_graph"] else "graph (per optimiser step"), agent.graph_status
            else:
                assert agent.graph_status["update"].startswith("graph (1 per iteration)"), agent.graph_status
            flat = torch.cat([p.detach().flatten() for p in agent.model.parameters()]).clone()
            assert torch.isfinite(flat).all()
            results.append((flat, float(agent.lr)))
            env.close()
        # captured collectives == eager collectives between per-step graphs == the plain single-rank run
        for other in results[1:]:
            assert torch.allclose(results[0][0], other[0], rtol=1e-5, atol=1e-6)
            assert results[0][1] == other[1]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


# ---------------------------------------------------------------- collective probe
class _StubEvent:
    """An event whose query() never (or after `after` polls) answers: the replayed collective that a peer never joins."""
    def __init__(self, after=None):
        self.after, self.polls = after, 0

    def query(self):
        self.polls += 1
        return self.after is not None and self.polls > self.after


def _fake_clock():
    t = {"now": 0.0}
    return (lambda: t["now"]), (lambda dt: t.__setitem__("now", t["now"] + dt))


def test_wait_bounded_times_out_on_an_event_that_never_completes():
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import wait_bounded
    clock, sleep = _fake_clock()
    ev = _StubEvent()
    assert wait_bounded(ev.query, 2.0, clock=clock, sleep=sleep, poll_s=0.25) is False
    assert 8 <= ev.polls <= 10 and clock() >= 2.0                 # it polled until the deadline, no longer
    ev = _StubEvent(after=3)
    assert wait_bounded(ev.query, 2.0, clock=clock, sleep=sleep, poll_s=0.25) is True and ev.polls == 4


def test_collective_probe_never_replays_unless_every_rank_captured():
    """a capture refused on ONE rank must keep every rank from replaying the probe graph."""
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import collective_probe_protocol
    calls = []

    def agree_other_rank_failed(ok):
        calls.append(("agree", ok))
        return False                                                # MIN over the ranks: a peer reported 0

    def replay(handle):
        calls.append(("replay",))
        return "ok"
    ok, why = collective_probe_protocol(lambda: (True, "captured", object()), replay, agree_other_rank_failed,
                                        abort=lambda m: calls.append(("abort", m)))
    assert ok is False and "another rank" in why
    assert calls == [("agree", True)]                               # no replay, one agreement, no abort
    # this rank's own refusal: it still joins the (one) agreement collective, so the ranks stay matched
    calls.clear()
    ok, why = collective_probe_protocol(lambda: (False, "capture refused: x", None), replay,
                                        lambda v: (calls.append(("agree", v)), False)[1], abort=lambda m: calls.append(("abort", m)))
    assert ok is False and why == "capture refused: x" and calls == [("agree", False)]


def test_collective_probe_timeout_branch_falls_back_or_aborts():
    from vine_robot_isaacgymenvs_amd.learning.a2c_continuous import collective_probe_protocol, wait_bounded
    clock, sleep = _fake_clock()

    def replay_hung(handle):                                        # the bounded wait on a stubbed event that never fires
        return "ok" if wait_bounded(_StubEvent().query, 5.0, clock=clock, sleep=sleep, poll_s=0.5) else "timeout"
    logs, aborts, agreed = [], [], []

    def agree(v):
        agreed.append(v)
        return v if len(agreed) == 1 else False                     # stage 1: all captured; stage 2: MIN is 0 (this rank timed out)
    ok, why = collective_probe_protocol(lambda: (True, "captured", object()), replay_hung, agree, abort=aborts.append,
                                        log=logs.append)
    assert ok is False and "timeout" in why and agreed == [True, False] and not aborts and logs
    # the ranks no longer answer after the hang: the process must end with a clear message, not wait forever
    agreed.clear()

    def agree_dead(v):
        agreed.append(v)
        return True if len(agreed) == 1 else None
    ok, why = collective_probe_protocol(lambda: (True, "captured", object()), replay_hung, agree_dead, abort=aborts.append,
                                        log=logs.append)
    assert ok is False and len(aborts) == 1 and "VINE_COLLECTIVE_IN_GRAPH=0" in aborts[0]
    # and the good case: two agreements, both true
    ok, why = collective_probe_protocol(lambda: (True, "captured", object()), lambda h: "ok", lambda v: v, abort=aborts.append)
    assert ok is True and len(aborts) == 1


def test_agree_bounded_on_cpu_single_rank_and_gloo_pair():
    """_agree_bounded answers like _capture_agreed when the collective completes (single rank: no collective at all)."""
    from tests.test_ppo import make_agent
    agent, _ = make_agent(num_envs=16, minibatch=64, seed=1, multi_gpu=False, max_epochs=1)
    assert agent._agree_bounded(True, 1.0) is True and agent._agree_bounded(False, 1.0) is False
    assert agent._collective_capture_ok() is False and agent.collective_capture["probe"] in ("no device", "disabled")
