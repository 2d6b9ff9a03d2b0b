"""CPU-only checks: the C-ABI library loads and exports every symbol include/sed_hip.h declares (no
compute calls), the ctypes prototypes cover the header, and the host-side mirror of the reference
API behaves like the reference (names, shapes, state_dict keys, seeded init, errors, metrics)."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "sed_hip.h")
LIB = os.path.join(ROOT, "soundeventdetection-pytorch_amd", "libsed_hip.so")
MAIN_CFG = [(32, 2), (64, 2), (128, 2), (128, 1)]
TINY_CFG = [(4, 2), (8, 2), (8, 2), (8, 1)]


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sed_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def sed():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return importlib.import_module("soundeventdetection-pytorch_amd")


def test_header_symbols_are_exported(sed):
    names = declared_functions()
    assert len(names) >= 30
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sed_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in sed_hip.h but not exported: {missing}"


def test_ctypes_prototypes_cover_header(sed):
    names = set(declared_functions())
    protos = set(sed._lib.PROTOTYPES)
    assert names == protos, (sorted(names - protos), sorted(protos - names))
    lib = sed._lib.lib()                       # dlopen + resolve; no kernel is launched
    assert lib.sed_abi_version() == 1
    assert lib.sed_conv_nparts(32, 6001, 64) == 1024
    assert lib.sed_conv_nparts(1, 4, 8) == 1
    assert lib.sed_logmel_ws_bytes(1, 1, 1024, 320) == 512 * 8


def test_argument_validation_without_gpu(sed):
    """Host-side checks fire before any launch and report through sed_last_error()."""
    lib = sed._lib.lib()
    rc = lib.sed_conv3x3_fwd(1, 0, 0, None, None, None, None, None, None, None, None, None, None, None, 1, 8, 64, 48, 32, None)
    assert rc != 0 and b"padded" in lib.sed_last_error()
    with pytest.raises(RuntimeError, match="padded"):
        sed._lib.check(rc, "conv3x3_fwd")
    rc = lib.sed_logmel_fwd(None, None, None, None, None, None, None, None, None, 1, 100, 1000, 10, 64, None)
    assert rc != 0 and b"power of two" in lib.sed_last_error()
    rc = lib.sed_adam_amsgrad_step(None, None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, None)
    assert rc != 0 and b"1-based" in lib.sed_last_error()


def test_state_dict_contract_and_seeded_init(sed):
    g = load_golden("g2_train_steps.npz")
    torch.manual_seed(0)
    m = sed.Cnn_AvgPooling(1, TINY_CFG)
    sd = m.state_dict()
    ref_keys = [k[len("tiny13.sd0."):] for k in g.files if k.startswith("tiny13.sd0.")]
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:                       # same RNG call order as the reference constructors
        assert sd[k].shape == g["tiny13.sd0." + k].shape
        assert np.array_equal(sd[k].numpy(), g["tiny13.sd0." + k]), k
    g3 = load_golden("g3_eval_forward.npz")
    m2 = sed.Cnn_AvgPooling(1, MAIN_CFG)
    for k, v in m2.state_dict().items():
        if "num_batches" not in k:
            assert tuple(v.shape) == g3["sd." + k].shape
    assert sum(p.numel() for p in m2.parameters()) == 582433
    assert sum(p.numel() for p in sed.Cnn_AvgPooling(1).parameters()) == 4686657    # class default config
    assert m2.num_pools == 3 and sed.Cnn_AvgPooling(1, [(8, 1), (8, 1)]).num_pools == 1


def test_no_cpu_path(sed):
    m = sed.Cnn_AvgPooling(1, TINY_CFG)
    with pytest.raises(RuntimeError, match="no CPU path|MI355X"):
        m(torch.zeros(1, 1, 16, 64))
    with pytest.raises(RuntimeError):
        sed.FusedTrainer(m, lr=1e-3)
    with pytest.raises(RuntimeError):
        m.conv_blocks[0](torch.zeros(1, 1, 16, 64))
    with pytest.raises(ValueError):
        sed.CnnEngine(1, [(8, 3)])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "soundeventdetection-pytorch_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").lower() or f.endswith(".md"), os.path.join(dp, f)


def test_metrics_match_reference_golden(sed):
    mu = importlib.import_module("soundeventdetection-pytorch_amd.utils.metric_utils")
    g = load_golden("g5_metrics.npz")
    for tag in ("rand", "no_gt", "all_gt", "len_mismatch", "k3", "edges"):
        r, p, ap = mu.calculate_metrics(g[f"{tag}.o"], g[f"{tag}.t"])
        assert np.array_equal(r, g[f"{tag}.recalls"]) and np.array_equal(p, g[f"{tag}.precisions"])
        assert ap == float(g[f"{tag}.AP"])
        assert np.array_equal(mu.f_score(p, r, 1), g[f"{tag}.f1"])
        assert np.array_equal(mu.f_score(p, r, 5), g[f"{tag}.f5"])     # swapped-argument convention
    rc, pr = mu.compute_recall_precision(g["crp.O"], g["crp.T"])
    assert rc == float(g["crp.recall"]) and pr == float(g["crp.prec"])
    assert len(mu.THRESHOLDS) == 21 and mu.THRESHOLDS[0] == 0.0


def test_validation_summary_uses_mean_curves(sed):
    tr = importlib.import_module("soundeventdetection-pytorch_amd.train")
    r = [np.linspace(1, 0, 21), np.linspace(1, 0.2, 21)]
    p = [np.linspace(0.1, 1, 21), np.linspace(0.3, 1, 21)]
    s = tr.summarize_validation([0.5, 0.7], r, p, [0.2, 0.4])
    rm, pm = np.mean(r, 0), np.mean(p, 0)
    f1 = 2 * pm * rm / (pm + rm + 1e-9)
    assert s["val_loss"] == pytest.approx(0.6) and s["AP"] == pytest.approx(0.3)
    assert s["max_f1"] == pytest.approx(f1.max())


def test_frontend_constants_match_oracle(sed):
    from oracle import frontend_oracle as FO
    pp = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.preprocess")
    sc = importlib.import_module("soundeventdetection-pytorch_amd.dataset.spectogram.spectogram_configs")
    assert (sc.REF_NATIVE.frame_size, sc.REF_NATIVE.hop_size, sc.REF_NATIVE.NFFT) == (31680, 15840, 32768)
    assert sc.REF_NATIVE.frames_per_second == 3 and sc.REF_NATIVE.train_crop_size == 30
    assert sc.BENCH.num_frames(60 * 32000) == 6001
    assert pp.MEL_FILTER_BANK_MATRIX.shape == (16385, 64)
    for c, o in ((sc.REF_NATIVE, FO.ref_native_config()), (sc.BENCH, FO.bench_config())):
        np.testing.assert_allclose(pp.mel_filter_bank(c), FO.mel_filter_bank_matrix(o), atol=1e-9)
        assert np.array_equal(pp.padded_window(c), FO.padded_window(o).astype(np.float32))
    x = np.random.default_rng(0).standard_normal((2, 9, 64)).astype(np.float32)
    m, s = pp.calculate_scalar_of_tensor(x)
    mo, so = FO.calculate_scalar_of_tensor(x)
    assert np.array_equal(m, mo) and np.array_equal(s, so)


def test_traffic_stamp_ignores_comments_and_white_space(tmp_path):
    """tools/csrc_sha.py (the stamp bench.py compares the PMC traffic table with) hashes CODE only: a comment-only or white-space-only
    edit of a kernel source or of the ABI header must leave `traffic_stale` alone, a code edit must flip it (task 7)."""
    import shutil
    from tools import csrc_sha as CS

    root = tmp_path / "tree"
    (root / "include").mkdir(parents=True)
    cs = root / "soundeventdetection-pytorch_amd" / "csrc"
    cs.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "include", "sed_hip.h"), root / "include" / "sed_hip.h")
    src = os.path.join(ROOT, "soundeventdetection-pytorch_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            shutil.copy(os.path.join(src, f), cs / f)
    base = CS.csrc_sha256(str(root))
    assert base == CS.csrc_sha256(ROOT)
    h = root / "include" / "sed_hip.h"
    text = h.read_text()
    h.write_text("/* a new caveat,\n * two lines */\n" + text.replace("\n", "   \n", 5) + "\n// trailing note\n")
    k = cs / "sed_eval.hip"
    ktext = k.read_text()
    k.write_text("// why this kernel exists\n" + ktext.replace("{", "{   // note", 1))
    mk = cs / "Makefile"
    mk.write_text("# a comment\n" + mk.read_text())
    assert CS.csrc_sha256(str(root)) == base, "comments / white space must not move the stamp"
    k.write_text(ktext.replace("{", "{ int sed_extra_ = 0; (void)sed_extra_;", 1))
    assert CS.csrc_sha256(str(root)) != base, "a code edit must move the stamp"
    # string literals are code: a '//' inside one is not a comment
    assert CS.strip_c('a = "x // y"; // z') == 'a = "x // y";'
    assert CS.strip_c("a /* b */ c\n\n  d") == "a c\nd"
