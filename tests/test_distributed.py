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
