"""The launcher against outputs of the reference's own legion_server.py (tests/golden/launcher.json,
generated in the build container by tests/golden/gen_launcher_golden.py)."""
import json
import os

from legion_amd import launcher

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "launcher.json")))


def test_meta_config_line_matches_reference():
    assert len(GOLD["meta_config"]) == 12
    for c in GOLD["meta_config"]:
        got = launcher.meta_config_line(c["dataset_path"], c["dataset_name"], c["train_batch_size"],
                                        c["cache_memory"], c["epoch"])
        assert got == c["line"]


def test_nvidia_matrix_parses_like_reference():
    for c in GOLD["topo"]:
        conns = launcher.parse_topo_output(c["text"])
        assert [list(x) for x in conns] == c["connections"], c["name"]
        assert launcher.largest_clique_size(conns) == c["clique_size"], c["name"]
        assert launcher.cache_agg_mode_for(launcher.largest_clique_size(conns)) == c["cache_agg_mode"]


def test_rocm_smi_matrix():
    text = ("============================ ROCm System Management Interface ============================\n"
            "=============================== Link Type between two GPUs ===============================\n"
            "       GPU0         GPU1         GPU2         GPU3         \n"
            "GPU0   0            XGMI         XGMI         PCIE         \n"
            "GPU1   XGMI         0            XGMI         PCIE         \n"
            "GPU2   XGMI         XGMI         0            PCIE         \n"
            "GPU3   PCIE         PCIE         PCIE         0            \n"
            "================================== End of ROCm SMI Log ===================================\n")
    conns = launcher.parse_topo_output(text)
    assert (0, 1) in conns and (3, 0) not in conns
    assert launcher.largest_clique_size(conns) == 3


def test_fanout_spellings():
    assert launcher.parse_fanout("[25,10]") == [25, 10]
    assert launcher.parse_fanout("15,10,5") == [15, 10, 5]
    assert launcher.parse_fanout(list("[25,10]")) == [25, 10]      # argparse type=list (legion_server.py:120)
    assert launcher.parse_fanout([25, 10]) == [25, 10]


def test_cli_defaults_match_reference():
    a = launcher.build_argparser().parse_args([])
    assert (a.dataset_path, a.dataset_name, a.train_batch_size, a.gpu_number, a.epoch, a.cache_memory, a.usenvlink) == \
        ("./dataset", "ukunion", 8000, 2, 2, 38000000, 1)


def test_failing_server_gives_nonzero_exit(tmp_path, monkeypatch):
    """A sampling_server that exits with EXIT_FAILURE (every HIP error, a missing meta_config ...) or dies by a
    signal must not look like success to whoever started legion_server.py."""
    import stat
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for body, want in (("exit 1", 1), ("exit 0", 0), ("kill -9 $$", 137)):
        fake = tmp_path / "fake_server"
        fake.write_text("#!/bin/sh\n" + body + "\n")
        fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
        monkeypatch.setattr(launcher, "server_binary", lambda f=str(fake): f)
        monkeypatch.chdir(tmp_path)
        args = launcher.build_argparser().parse_args(["--dataset_name", "products", "--usenvlink", "0", "--fanout", "25,10"])
        assert launcher.Run(args) == want
    # and end to end through legion_server.py's sys.exit
    code = ("import sys; sys.path.insert(0, %r); from legion_amd import launcher; "
            "launcher.server_binary = lambda: %r; sys.exit(launcher.main(['--dataset_name','products','--usenvlink','0']))")
    fake.write_text("#!/bin/sh\nexit 1\n")
    rc = subprocess.call([sys.executable, "-c", code % (root, str(fake))], cwd=tmp_path)
    assert rc == 1
