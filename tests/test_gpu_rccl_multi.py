"""-m gpu: the multi-GPU layer over the REAL wire -- one rank per GPU, backend "nccl" (= RCCL over xGMI).

The parity gate for BASELINE config C4's exchange: the first multi-rank RCCL run of this library on any node must be a
parity test, not the timed bench.  On a box with >= 2 GPUs it spawns N = min(8, device_count) ranks (and a 2-rank world,
where the planner would pick another strategy) and compares distributed_inner_join (key shuffle), fused_inner_join,
broadcast_inner_join -- each through global_ids() -- and distributed_group_by (sum / min / max / count / avg) with the
expected pairs / the oracle / pandas on 1e6-row shards, then on uneven shards incl. a rank with NO probe rows, a rank with two,
and a rank without build rows.  The workers and checkers are the ones tests/test_gpu_multirank_one_gpu.py runs over gloo on
one GPU (tests/multirank_common.py); only the process-group set-up differs.

The 1-GPU boxes cannot run that (RCCL refuses two ranks on one device): there the same worker runs as a ONE-rank RCCL world, so
that the test code itself -- device selection, un-staged collectives, the checkers at the "rccl" shard size -- is exercised
before a multi-GPU node ever sees it."""
import pytest
import torch
import torch.multiprocessing as mp

from multirank_common import _free_port, _gather_worker, _multikey_worker, _uneven_worker, _worker, check_gather, check_join_and_group_by, check_multikey, check_uneven

pytestmark = pytest.mark.gpu

NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0


def _run_ranks(target, world, extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + extra + (q, "nccl")) for r in range(world)]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=900) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=180)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    return results


@pytest.mark.timeout(1200)
def test_one_rank_rccl_world_runs_the_same_workers():
    check_join_and_group_by(1, "rccl", _run_ranks(_worker, 1, ("rccl",)))


@pytest.mark.timeout(1800)
@pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
@pytest.mark.parametrize("world", sorted({2, min(8, max(NGPU, 2))}))
def test_multi_rank_rccl_parity(world):
    check_join_and_group_by(world, "rccl", _run_ranks(_worker, world, ("rccl",)))


@pytest.mark.timeout(1800)
@pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
def test_multi_rank_rccl_uneven_and_empty_shards():
    world = min(8, NGPU)
    check_uneven(world, _run_ranks(_uneven_worker, world, ()))


@pytest.mark.timeout(1200)
def test_one_rank_rccl_world_multi_key_group_by():
    """gdf_amd_dist_group_by_multi through the RCCL transport's all_to_all_v at world 1 (the exchange is a self-copy): the worker and
    the checker a multi-GPU node will run"""
    check_multikey(1, _run_ranks(_multikey_worker, 1, ()))


@pytest.mark.timeout(1800)
@pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
def test_multi_rank_rccl_multi_key_group_by():
    """several key columns + validity masks across real ranks: grouped ncclSend / ncclRecv with exact sizes (all_to_all_v); the rank
    without rows takes part in every collective"""
    world = min(8, NGPU)
    check_multikey(world, _run_ranks(_multikey_worker, world, ()))



@pytest.mark.timeout(1200)
def test_one_rank_rccl_world_materialisation():
    """gdf_amd_dist_gather through the RCCL transport's all_to_all_v at world 1 (requests and answers are self-copies): the worker and the
    checker a multi-GPU node will run"""
    check_gather(1, _run_ranks(_gather_worker, 1, ()))


@pytest.mark.timeout(1800)
@pytest.mark.skipif(NGPU < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
def test_multi_rank_rccl_materialisation():
    """the distributed result_cols step across real ranks: local rows out, values and valid flags back, exact sizes both ways"""
    world = min(8, NGPU)
    check_gather(world, _run_ranks(_gather_worker, world, ()))
