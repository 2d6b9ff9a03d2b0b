"""Host-side pieces of the M5 path that need no GPU: state_dict contract, seeded init, gradient buckets,
waveform dataset protocol (reference: models/waveform_models.py, dataset/waveform/waveform_dataset.py)."""
import importlib
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_m5.npz"))


def _pkg():
    return importlib.import_module("soundeventdetection-pytorch_amd")


def test_state_dict_keys_shapes_and_seeded_init():
    sed = _pkg()
    torch.manual_seed(0)
    m = sed.M5(1)
    sd = m.state_dict()
    ref = {k[4:]: G[k] for k in G.files if k.startswith("sd0.")}
    assert list(sd.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
        assert np.array_equal(sd[k].numpy(), v), k        # same RNG draws as nn.Conv1d / nn.Linear
    assert sum(p.numel() for p in m.parameters()) == 426369


def test_gradient_buckets_follow_backward_order():
    sed = _pkg()
    m = sed.M5(1)
    flat = sed.train.FlatParams(m)
    assert [k for k, _, _ in flat.groups] == ["fc", "conv_block5", "conv_block4", "conv_block3", "conv_block2", "conv_block1"]
    # merged into two all-reduce buckets in backward completion order: head >= 85 % of the elements, small tail
    keys = [list(k) for k, _, _ in flat.buckets]
    assert len(keys) == 2 and sum(keys, []) == [k for k, _, _ in flat.groups]
    ends = [e for _, _, e in flat.buckets]
    assert max(ends) == flat.numel and min(s for _, s, _ in flat.buckets) == 0


def test_cpu_forward_raises():
    sed = _pkg()
    m = sed.M5(1)
    try:
        m(torch.zeros(8, 1, 2048))
    except RuntimeError as e:
        assert "MI355X" in str(e)
    else:
        raise AssertionError("a CPU forward must fail loudly")


def test_waveform_dataset_protocol_and_labels():
    wd = importlib.import_module("soundeventdetection-pytorch_amd.dataset.waveform.waveform_dataset")
    cfg = importlib.import_module("soundeventdetection-pytorch_amd.dataset.waveform.waveform_configs")
    assert cfg.frame_size == 31680 and cfg.hop_size == 15840 and cfg.frames_per_second == 3
    np.random.seed(0)
    items, waves = wd.synthetic_waveform_task(n_files=3, seconds=6.0, seed=1)
    ds = wd.WaveformDataset(items, val_descriptor="val_", waveforms=waves)
    x, y = ds[3]
    assert x.shape == (1, cfg.frame_size) and isinstance(bool(y), bool)
    # start-index labels (waveform_dataset.py:34-44): a frame starting right at an event start is covered
    (path, st, en, name) = [t for t in items if "train" in t[0]][0]
    lab = wd.get_start_indices_labesl(waves[path].shape[1], st, en)
    s0 = int(st[0] * cfg.working_sample_rate)
    assert lab[s0 - int(cfg.frame_size * 0.2)] == 1 and lab[max(0, s0 - cfg.frame_size)] == 0
    frames, labels, fname = next(iter(ds.get_validation_sampler(3)))
    assert frames.shape[1:] == (1, cfg.frame_size) and labels.shape[0] == frames.shape[0]
    # hop-size split (:9-31): one frame per hop, centred
    n_expected = len(np.arange(cfg.frame_size // 2, int(6.0 * cfg.working_sample_rate) - cfg.frame_size // 2 + 1, cfg.hop_size))
    assert frames.shape[0] == n_expected


def test_waveform_loader_refuses_an_epoch_without_batches():
    """Fewer training frames than one global batch: the reference's DataLoader would yield a short batch; this loader
    drops ragged tails (multiples of 8 frames per rank), so an empty epoch must be an error, not an endless loop."""
    wd = importlib.import_module("soundeventdetection-pytorch_amd.dataset.waveform.waveform_dataset")

    class Tiny:
        def __len__(self):
            return 100

    loader = wd.WaveformBatchLoader(Tiny(), 64, rank=0, world_size=2, device="cpu")
    assert len(loader) == 0
    try:
        next(iter(loader))
    except ValueError as e:
        assert "batch" in str(e)
    else:
        raise AssertionError("an epoch without batches must raise")
