"""The multi-GPU product entry `python -m holo_diffusion_amd.generate` (generate_samples.py:37-51,141-149 of the
reference as an OmegaConf-free key=value CLI): argument parsing on the host, and the whole entry at world size 2 over
gloo with a stand-in model (tests/support/generate_cli_stub.py) - on the GPU node the same entry runs over RCCL."""
import os
import subprocess
import sys

import pytest
import torch
import yaml

from holo_diffusion_amd import generate as gen

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_cli_mirrors_generate_samples_arguments():
    cfg = gen.parse_cli(["exp_dir=/x/y", "n_eval_cameras=40", "render_size=[64,48]", "num_samples=8", "seed=11",
                         "progressive_sampling_steps_per_render=2", "up=[0.0,-1.0,0.0]"])
    assert cfg["exp_dir"] == "/x/y" and cfg["n_eval_cameras"] == 40 and cfg["render_size"] == (64, 48)
    assert cfg["num_samples"] == 8 and cfg["seed"] == 11 and cfg["up"] == (0.0, -1.0, 0.0)
    assert cfg["camera_path"] == "simple_360" and cfg["video_size"] == (256, 256)  # defaults of generate_samples.py:37-51
    assert gen.parse_cli([])["n_eval_cameras"] == 75 and gen.parse_cli([])["up"] == gen.CANONICAL_CO3D_UP_AXIS
    with pytest.raises(SystemExit):
        gen.parse_cli(["no_such_key=1"])
    with pytest.raises(SystemExit):
        gen.parse_cli(["positional"])
    with pytest.raises(SystemExit):
        gen.main([])  # exp_dir is required


def _experiment(tmp_path):
    from tests.test_checkpoint_loading import _expconfig
    d = tmp_path / "exp"
    d.mkdir()
    with open(d / "expconfig.yaml", "w") as f:
        yaml.safe_dump(_expconfig(), f)
    return str(d)


def _run(cmd, env):
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    return res.stdout


def test_generate_cli_world2_gloo_equals_single_process(tmp_path):
    exp = _experiment(tmp_path)
    script = os.path.join(REPO, "tests", "support", "generate_cli_stub.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    args = ["num_samples=3", "n_eval_cameras=2", "render_size=[12,10]", "seed=5"]
    port = 29600 + os.getpid() % 1500
    out2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                 "127.0.0.1", "--master-port", str(port), script, f"exp_dir={exp}",
                 f"output_directory={tmp_path / 'two'}"] + args, env)
    assert "on 2 rank(s), backend gloo" in out2
    out1 = _run([sys.executable, script, f"exp_dir={exp}", f"output_directory={tmp_path / 'one'}"] + args, env)
    assert "on 1 rank(s)" in out1
    for i in range(3):  # per-sample seeds: a sample is the same whichever rank drew it
        a = torch.load(str(tmp_path / "two" / f"sample_{i:05d}_frames.pt"))
        b = torch.load(str(tmp_path / "one" / f"sample_{i:05d}_frames.pt"))
        assert a["images_render"].shape == (2, 3, 10, 12)
        for k in ("images_render", "depths_render", "masks_render"):
            assert torch.equal(a[k], b[k]), (i, k)
    a0 = torch.load(str(tmp_path / "two" / "sample_00000_frames.pt"))["images_render"]
    a1 = torch.load(str(tmp_path / "two" / "sample_00001_frames.pt"))["images_render"]
    assert not torch.equal(a0, a1)


_HAVE_EMU = os.path.isfile(os.path.join(REPO, "tests", "emu", "libholo_emu.so"))


@pytest.mark.skipif(not _HAVE_EMU, reason="needs the host emulation library (`make -C holo_diffusion_amd/csrc emu`)")
def test_generate_cli_world2_gloo_on_the_emulated_kernels_short(tmp_path):
    """The multi-rank product entry with the REAL model on the host emulation of the kernels, cheap enough for every CPU
    suite: chains subsampled to 2 of their 20 steps (the launcher's HOLO_TEST_MAX_ITER), 2 samples over 2 ranks, 2 cameras at
    8 x 8 - equal, bit for bit, to the single-process run."""
    _emulated_cli_case(tmp_path, max_iter=2, tiny=True)


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("HOLO_TEST_EMU_SLOW") != "1" or not _HAVE_EMU,
                    reason="opt-in (HOLO_TEST_EMU_SLOW=1 + `make -C holo_diffusion_amd/csrc emu`): ~15 minutes of host emulation")
def test_generate_cli_world2_gloo_on_the_emulated_kernels(tmp_path):
    """The same with the whole 20-step schedule."""
    _emulated_cli_case(tmp_path, max_iter=None)


def _emulated_cli_case(tmp_path, max_iter, tiny=False):
    """The same entry with the REAL model on the host emulation of the kernels (tests/support/generate_cli_emu.py): a
    20-step DDPM schedule on an 8^3 x 16 grid, 2 samples sharded over 2 ranks, frames gathered over gloo - equal, bit for bit,
    to the single-process run of the same samples (per-sample seeds seed + i), and different between samples."""
    from holo_diffusion_amd import checkpoint as ck
    import holo_diffusion_amd as hda
    from tests.test_checkpoint_loading import _expconfig, _reference_like_state
    d = tmp_path / "exp"
    d.mkdir()
    cfg = _expconfig(resol=4 if tiny else 8, feat=16, mc=32)
    margs = cfg["model_factory_ImplicitronModelFactory_args"]["model_HoloDiffusionModel_args"]
    margs["diffusion_args"]["num_steps"] = 20  # (the shortest schedule whose scaled linear betas stay <= 1)
    if tiny:  # one level, one ResBlock per side, the middle block's attention at T = 64: a forward of a dozen small launches
        margs["net_3d_SimpleUnet3D_args"].update(num_res_blocks=1, channel_mult=[1], attention_resolutions=[])
    with open(d / "expconfig.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    kw, _ = ck.model_args_from_expconfig(ck.read_expconfig(str(d))[0])
    torch.save(_reference_like_state(hda.HoloDiffusionModel(**kw), 11), str(d / "model_epoch_00000001.pth"))
    script = os.path.join(REPO, "tests", "support", "generate_cli_emu.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    if max_iter:
        env["HOLO_TEST_MAX_ITER"] = str(max_iter)
    args = ["num_samples=2", "n_eval_cameras=2", "render_size=[8,8]", "seed=5"]
    port = 29700 + os.getpid() % 1500 + (0 if max_iter else 3)

    def run(cmd):
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=3000, env=env)
        assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
        return res.stdout

    out2 = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), script, f"exp_dir={d}", f"output_directory={tmp_path / 'two'}"] + args)
    assert "on 2 rank(s), backend gloo" in out2
    out1 = run([sys.executable, script, f"exp_dir={d}", f"output_directory={tmp_path / 'one'}"] + args)
    assert "on 1 rank(s)" in out1
    frames = []
    for i in range(2):
        a = torch.load(str(tmp_path / "two" / f"sample_{i:05d}_frames.pt"))
        b = torch.load(str(tmp_path / "one" / f"sample_{i:05d}_frames.pt"))
        for k in ("images_render", "depths_render", "masks_render"):
            assert torch.equal(a[k], b[k]), (i, k)
        frames.append(a["images_render"])
    assert not torch.equal(frames[0], frames[1])
