"""bench.py's command-line contract, checked without a GPU: a plain `python bench.py --gpus N` (how the driver invokes it,
no torchrun around it) must turn itself into N ranks under torch.distributed.run on 127.0.0.1 -- round 1 asserted
WORLD_SIZE == --gpus and exited (missing item 1)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_world_size_self_launches(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    captured = {}

    def fake_execv(path, argv):
        captured["path"], captured["argv"] = path, list(argv)
        raise SystemExit(0)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit):
        bench.main()
    argv = captured["argv"]
    assert captured["path"] == sys.executable and argv[0] == sys.executable
    assert argv[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in argv and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(argv[argv.index("--master-port") + 1]) < 65536
    script = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[script + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]        # the user's flags travel unchanged
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"                            # RCCL needs dmabuf IPC on this host driver


def test_build_id_tracks_the_kernel_sources():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.library_build_id()
    assert len(a) == 16 and a == bench.library_build_id()
    # the newest committed PMC summary is either for THIS build or refused -- never silently stale
    traffic, src = bench.pmc_traffic("jk_scatter1", 2.0)
    assert traffic is None or "refused" not in (src or "")


def test_committed_pmc_summary_matches_this_build():
    """roofline.traffic of the driver's bench line comes from profiles/*_pmc_hbm.json and is refused when that file measured other kernel
    sources.  A tree whose csrc/ changed after the last collection skips here (a reminder, not a failure): re-run tools/gpu/r6_final.sh
    and commit its pmc_hbm.json."""
    sys.path.insert(0, ROOT)
    import bench
    import pytest
    traffic, src = bench.pmc_traffic(None, 1.0)
    if traffic is None:
        pytest.skip(f"no PMC summary for build {bench.library_build_id()}: {src}")
    assert 3.0e10 < traffic < 6.0e10, (traffic, src)          # one C3 join moves ~46 GB
