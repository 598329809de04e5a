"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute calls), the flag surface and the state-dict keys equal the reference's
(golden data), and the product path refuses to run without a GPU instead of falling back."""
import json
import os
import re
import sys

import pytest
import torch

from conftest import GOLDEN, ROOT, load_golden


def test_library_exports_every_declared_symbol():
    from idee_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "idee_hip.h")).read()
    declared = set(re.findall(r"\b(idee_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.idee_version().decode().startswith("idee_hip")
    # a library built with kernel timing-variant macros (IDEE_EXTRA_FLAGS=-DIDEE_DBG_SKIP=..: loads / MFMAs / stores skipped) says
    # so in its version string and is refused here
    assert "TIMING-VARIANT" not in lib.idee_version().decode(), lib.idee_version().decode()
    assert _lib.last_error() == ""


def test_option_table_set_get_restore():
    """idee_set_option / idee_get_option (no launch involved): values stick, "" / None reset, unknown names are refused."""
    from idee_amd import _lib
    before = _lib.get_option("IDEE_SWIN_MATH")
    with _lib.options(IDEE_SWIN_MATH="f22", IDEE_CONV_HRUN=8):
        assert _lib.get_option("IDEE_SWIN_MATH") == "f22" and _lib.get_option("IDEE_CONV_HRUN") == "8"
        _lib.set_option("IDEE_SWIN_MATH", None)
        assert _lib.get_option("IDEE_SWIN_MATH") == ""
    assert _lib.get_option("IDEE_SWIN_MATH") == before and _lib.get_option("IDEE_CONV_HRUN") == ""
    with pytest.raises(_lib.IdeeHipError, match="unknown option"):
        _lib.set_option("IDEE_NO_SUCH_SWITCH", "1")
    with pytest.raises(_lib.IdeeHipError, match="too long"):
        _lib.set_option("IDEE_SWIN_MATH", "x" * 40)


def test_config_surface_equals_reference():
    import argparse
    from idee_amd import config
    ref = json.load(open(os.path.join(GOLDEN, "config_defaults.json")))
    mine = json.loads(json.dumps(vars(config.add_all_arguments(argparse.ArgumentParser()).parse_args([])), default=list))
    assert mine == ref
    c = config.read_arguments(train=False, argv=[])
    assert c.phase == "test" and c.encoder == "Mamba"


@pytest.mark.parametrize("tag,enc", [("swin_v6_8x8_c2", "Swin_3D"), ("swin_v2_14x18", "Swin_3D")])
def test_state_dict_keys_shapes_roundtrip(tag, enc):
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    a, sd, _ = load_golden("e2e_" + tag)
    m = VQ_model(default_config(encoder=enc, in_channels=int(a["cfg_C"]), in_channels_dynamic=int(a["cfg_V"])))
    msd = m.state_dict()
    assert list(msd) == list(sd), "state-dict keys must come in the reference's registration order"
    assert all(tuple(msd[k].shape) == tuple(sd[k].shape) and msd[k].dtype == sd[k].dtype for k in sd)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    bad = dict(sd); bad["encoder.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
    # the reference's count_parameters(model.encoder / .cls / .vq) sums (train_synthetic.py:108-111)
    n = lambda mod: sum(p.numel() for p in mod.parameters() if p.requires_grad)
    assert n(m) == sum(v.numel() for k, v in sd.items() if v.is_floating_point())


def test_unknown_encoder_and_mamba_rejected():
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    with pytest.raises(NotImplementedError):
        VQ_model(default_config())               # reference default 'Mamba' is outside the path
    with pytest.raises(NotImplementedError):
        VQ_model(default_config(encoder="Nope"))


def test_no_cpu_fallback():
    """A CPU tensor must raise (there is no CPU/eager path), and nothing under idee_amd imports the oracle."""
    from idee_amd import _lib
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    m = VQ_model(default_config(encoder="Swin_3D", in_channels=1, in_channels_dynamic=2))
    with pytest.raises(_lib.IdeeHipError):
        m(torch.randn(1, 2, 1, 8, 8, 8))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "idee_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_window_clamp_matches_oracle():
    from idee_amd.ops import clamp_window
    from oracle import idee_oracle as O
    for ext in [(8, 200, 200), (8, 3, 5), (16, 4, 6), (1, 1, 1)]:
        for w, s in [((2, 4, 4), (1, 2, 2)), ((8, 1, 1), (4, 0, 0))]:
            assert clamp_window(ext, w, s) == O.get_window_size(ext, w, s)


def test_cosine_lr_schedule_known_answers():
    """timm CosineLRScheduler semantics at the reference's settings (config.py:121-126, utils_train.py:156-167):
    closed-form values (parity unpinned: timm is absent)."""
    import math
    from idee_amd.harness import CosineLRScheduler

    class Opt:
        param_groups = [dict(lr=1e-3)]
    o = Opt()
    s = CosineLRScheduler(o, t_initial=100, lr_min=1e-5, warmup_t=2, warmup_lr_init=1e-6)
    assert o.param_groups[0]["lr"] == 1e-6                      # set at construction
    s.step_update(1); assert abs(o.param_groups[0]["lr"] - (1e-6 + (1e-3 - 1e-6) / 2)) < 1e-15
    s.step_update(2); assert abs(o.param_groups[0]["lr"] - (1e-5 + 0.5 * (1e-3 - 1e-5) * (1 + math.cos(math.pi * 2 / 100)))) < 1e-15
    s.step_update(50); assert abs(o.param_groups[0]["lr"] - (1e-5 + 0.5 * (1e-3 - 1e-5))) < 1e-12
    s.step_update(100); assert o.param_groups[0]["lr"] == 1e-5   # past the single cycle


def test_step_lr_schedule_known_answers():
    """timm StepLRScheduler semantics at the reference's settings (utils_train.py:146-154; t_in_epochs=True): `step(epoch)`
    applies warm-up then base * rate ** (t // decay_t); `step_update` is ignored, which is what the reference's drivers
    call (so their learning rate stays at warmup_lr_init)."""
    from idee_amd.harness import StepLRScheduler, get_learning_scheduler
    from idee_amd.config import default_config

    class Opt:
        param_groups = [dict(lr=1e-3)]
    o = Opt()
    s = StepLRScheduler(o, decay_t=20, decay_rate=0.9, warmup_t=2, warmup_lr_init=1e-6)
    assert o.param_groups[0]["lr"] == 1e-6
    s.step_update(50); assert o.param_groups[0]["lr"] == 1e-6          # ignored (t_in_epochs)
    s.step(1); assert abs(o.param_groups[0]["lr"] - (1e-6 + (1e-3 - 1e-6) / 2)) < 1e-15
    s.step(2); assert o.param_groups[0]["lr"] == 1e-3
    s.step(19); assert o.param_groups[0]["lr"] == 1e-3
    s.step(20); assert abs(o.param_groups[0]["lr"] - 0.9e-3) < 1e-15
    s.step(45); assert abs(o.param_groups[0]["lr"] - 1e-3 * 0.9 ** 2) < 1e-15
    o2 = Opt(); o2.param_groups = [dict(lr=1e-3)]
    cfg = default_config(lr_scheduler="step")
    assert isinstance(get_learning_scheduler(o2, cfg), StepLRScheduler)
    cfg.lr_scheduler = "linear"
    with pytest.raises(ValueError):
        get_learning_scheduler(o2, cfg)


def test_adam_state_dict_interchanges_with_torch_adam():
    """idee_amd.optim.Adam speaks torch.optim.Adam's state-dict layout both ways (checkpoint interchange with the
    reference's `optimizer_state_dict`, utils_train.py:562-584) and refuses a parameter without a gradient."""
    from idee_amd import optim
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    ref = torch.optim.Adam(lin.parameters(), lr=2e-3, weight_decay=0.003, betas=(0.8, 0.99))
    lin(torch.randn(4, 5)).sum().backward()
    ref.step()
    ours = optim.Adam(lin.parameters(), lr=1e-3)
    ours.load_state_dict(ref.state_dict())
    assert ours.step_count == 1 and ours.param_groups[0]["lr"] == 2e-3 and tuple(ours.param_groups[0]["betas"]) == (0.8, 0.99)
    off = 0
    for i, p in enumerate(ours.params):
        st = ref.state_dict()["state"][i]
        assert torch.equal(ours.exp_avg[off:off + p.numel()].view(p.shape), st["exp_avg"])
        assert torch.equal(ours.exp_avg_sq[off:off + p.numel()].view(p.shape), st["exp_avg_sq"])
        off += p.numel()
    sd = ours.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    lin2 = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    back = torch.optim.Adam(lin2.parameters())
    back.load_state_dict(sd)                                           # torch accepts it
    assert back.state_dict()["state"][2]["exp_avg"].shape == (2, 3) and float(back.state_dict()["state"][0]["step"]) == 1.0
    # a parameter without a gradient: torch would skip it; the flat step refuses instead of decaying it silently
    ours.zero_grad(set_to_none=True)
    lin[0](torch.randn(4, 5)).sum().backward()
    with pytest.raises(RuntimeError, match="no gradient"):
        ours.gather_grads()


@pytest.mark.parametrize("tag,enc", [("swin_v6_8x8_c2", "Swin_3D"), ("cnn_v2_12x16", "CNN_3D")])
def test_optimizer_state_interchanges_with_reference_shaped_torch_adam(tag, enc):
    """`optimizer_state_dict` of a checkpoint (utils_train.py:562-584) is indexed by the REFERENCE model's parameters — one per
    variable (~240 for Swin), not this code base's V-stacked ones.  A torch.optim.Adam built over a reference-shaped parameter
    list (the golden state dict's float entries, in its order) must load what idee_amd.optim.Adam writes, and the reverse,
    moment for moment."""
    from idee_amd import optim
    from idee_amd.config import default_config
    from idee_amd.models.build import VQ_model
    from idee_amd.models._stacked import reference_param_slices
    a, sd, _ = load_golden("e2e_" + tag)
    V = int(a["cfg_V"])
    m = VQ_model(default_config(encoder=enc, in_channels=int(a["cfg_C"]), in_channels_dynamic=V))
    m.load_state_dict(sd, strict=True)
    ref_keys = [k for k, v in sd.items() if v.is_floating_point()]
    slices = reference_param_slices(m)
    assert [k for k, _, _, _ in slices] == ref_keys and all(tuple(sd[k].shape) == r for k, _, _, r in slices)
    # the reference side: torch Adam over reference-shaped parameters, two steps with seeded gradients
    g = torch.Generator().manual_seed(5)
    ref_params = [torch.nn.Parameter(sd[k].clone()) for k in ref_keys]
    ref = torch.optim.Adam(ref_params, lr=2e-3, weight_decay=0.003, betas=(0.8, 0.99))
    for _ in range(2):
        for p in ref_params:
            p.grad = torch.randn(p.shape, generator=g)
        ref.step()
    ours = optim.Adam(m.parameters(), lr=1e-3, reference_slices=slices)
    assert len(ours.params) < len(ref_params)                      # V-stacked here, per variable there
    ours.load_state_dict(ref.state_dict())
    assert ours.step_count == 2 and ours.param_groups[0]["lr"] == 2e-3
    where = {id(p): (lo, hi, shape) for p, (lo, hi, shape) in zip(ours.params, ours._spans())}
    for i, (key, p, idx, rshape) in enumerate(slices):
        lo, hi, shape = where[id(p)]
        for buf, name in ((ours.exp_avg, "exp_avg"), (ours.exp_avg_sq, "exp_avg_sq")):
            got = buf[lo:hi].view(shape)
            got = (got if idx is None else got[idx]).reshape(rshape)
            assert torch.equal(got, ref.state[ref_params[i]][name]), key
    # and back: a fresh reference-shaped torch Adam accepts our state dict (torch validates the group sizes)
    back_params = [torch.nn.Parameter(sd[k].clone()) for k in ref_keys]
    back = torch.optim.Adam(back_params)
    back.load_state_dict(ours.state_dict())
    for p0, p1 in zip(ref_params, back_params):
        assert torch.equal(back.state[p1]["exp_avg"], ref.state[p0]["exp_avg"])
        assert torch.equal(back.state[p1]["exp_avg_sq"], ref.state[p0]["exp_avg_sq"])
        assert float(back.state[p1]["step"]) == 2.0
    assert back.param_groups[0]["lr"] == 2e-3 and tuple(back.param_groups[0]["betas"]) == (0.8, 0.99)
    # a checkpoint indexed for another layout is refused with a clear error, before any moment is touched
    other = optim.Adam(m.parameters(), lr=1e-3)                    # own (stacked) indexing
    with pytest.raises(ValueError, match="indexes"):
        other.load_state_dict(ref.state_dict())


def test_launch_ranks_stops_siblings_on_first_failure(tmp_path, monkeypatch):
    """bench.launch_ranks polls its children: when one rank dies (here: rank 1 exits 3 at once) the others — which would sit
    in the rendezvous until the driver's limit — are terminated and the launcher returns the failing status promptly."""
    import argparse
    import time
    import bench
    monkeypatch.setenv("IDEE_FORCE_DEVICE", "0")          # the children here never touch a GPU: skip the visible-GPU count
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                     "time.sleep(120)\n")
    t0 = time.time()
    rc = bench.launch_ranks(argparse.Namespace(gpus=3), argv=[sys.executable, str(child)], poll_s=0.05)
    assert rc == 3 and time.time() - t0 < 30
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n")
    assert bench.launch_ranks(argparse.Namespace(gpus=2), argv=[sys.executable, str(ok)], poll_s=0.05) == 0


def test_bench_refuses_rank_count_mismatch():
    """`bench.py --gpus N` never reports a line for another rank count: a torchrun-style environment with a different
    WORLD_SIZE is refused before any GPU work, and the self-launch path refuses when fewer GPUs are visible."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IDEE_FORCE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        # either the launcher counts the GPUs (KFD topology in sysfs, no HIP call) and refuses, or — where sysfs shows no KFD —
        # the first child that finds no device exits non-zero and the launcher stops its siblings; never a JSON line
        assert r.returncode != 0 and ("refusing" in r.stderr or "stopping the other ranks" in r.stderr) and '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout) and '"metric"' not in r.stdout


def test_lfq_indices_to_codes_video_map_and_qkv_bias_false_keys():
    """LFQ.indices_to_codes on an index map with >= 3 dims returns channels-first codes (LFQ.py:152-181; golden), and a
    Swin_3D built with qkv_bias=False has no attn.qkv.bias keys (nn.Linear(bias=False), Swin_3D.py:137)."""
    from idee_amd.models.codebook.LFQ import LFQ
    from idee_amd.models.encoder.Swin_3D import Swin_3D
    a, _, _ = load_golden("misc")
    q = LFQ(dim=16, codebook_size=4, entropy_loss_weight=0.1, diversity_gamma=0.1, commitment_loss_weight=3.0)
    q.load_state_dict({k[7:]: v for k, v in a.items() if k.startswith("lfq/sd/")}, strict=True)
    codes = q.indices_to_codes(a["lfq/indices"])
    assert tuple(codes.shape) == (2, 16, 3, 4, 5)
    torch.testing.assert_close(codes, a["lfq/codes"], atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(q.indices_to_codes(a["lfq/indices"][0, 0, 0]), a["lfq/codes_flat"], atol=1e-6, rtol=1e-6)
    enc = Swin_3D(in_vars=2, in_chans=1, qkv_bias=False)
    keys = set(enc.state_dict())
    assert not any(k.endswith("attn.qkv.bias") for k in keys) and any(k.endswith("attn.qkv.weight") for k in keys)
    assert not any(n.endswith("qkv_b") for n, _ in enc.named_parameters())
