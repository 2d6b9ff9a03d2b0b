"""Training CLI: every flag of /root/reference/main.py:89-117 with the same defaults, driving the
MI355X spectrogram path.  Run as `python -m sed_amd.main ...` from the repository root (or under
torch.distributed.run for data parallel: one process per GPU, RCCL gradient all-reduce).

Beyond the reference: `--dataset_name synthetic` (a seeded in-memory dataset; TAU / FilmClap audio
cannot be fetched on a box without network) and `--precision`.  `--train_features Waveform` (the
M5 model) trains through the same loop; `--dataset_name synthetic` gives a seeded
stand-in task for either feature type."""
from __future__ import annotations

import argparse
import os

import torch

from .train import train
from .utils.common import WeightedBCE


def build_parser():
    p = argparse.ArgumentParser(description="SED training on MI355X")
    # Training
    p.add_argument("--dataset_dir", type=str, default="../data", help="Directory of dataset.")
    p.add_argument("--dataset_name", type=str, default="FilmClap", help="FilmClap, TAU or synthetic")
    p.add_argument("--train_features", type=str, default="Waveform", help="Spectogram or Waveform")
    # Spectogram only arguments
    p.add_argument("--preprocess_mode", type=str, default="logMel",
                   help="logMel or Complex; relevant only for Spectogram features")
    p.add_argument("--force_preprocess", action="store_true", default=False,
                   help="relevant only for Spectogram features")
    # Train
    p.add_argument("--outputs_root", type=str, default="training_dir")
    p.add_argument("--ckpt", type=str, default="")
    p.add_argument("--val_descriptor", default=0.2,
                   help="float for percentage string for specifying fold substring")
    p.add_argument("--train_tag", type=str, default="")
    # Training tricks
    p.add_argument("--augment_data", action="store_true", default=False)
    p.add_argument("--balance_classes", action="store_true", default=False,
                   help="Whether to make sure there is same number of samples with and without events")
    p.add_argument("--recall_priority", type=float, default=5, help="priority factor for the bce loss")
    # Hyper parameters
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--lr", type=float, default=0.000001)
    p.add_argument("--num_train_steps", type=int, default=100000)
    p.add_argument("--log_freq", type=int, default=5000)
    # Infrastructure
    p.add_argument("--device", default="cuda:0", type=str)
    p.add_argument("--num_workers", default=12, type=int,
                   help="accepted for compatibility: batches are assembled on the GPU, there are no workers")
    # this build only
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "f16x3", "bf16x3"])
    return p


def _val_descriptor(v):
    """argparse hands a string for an explicit --val_descriptor; the reference's float default selects
    a percentage split: accept '0.2' as a float as well (main.py:102)."""
    if isinstance(v, float):
        return v
    try:
        return float(v)
    except ValueError:
        return v


def get_spectogram_dataset_model_and_criterion(args, device):
    """main.py:10-46."""
    from .dataset.spectogram import spectogram_configs as cfgs
    from .dataset.spectogram.spectograms_dataset import (SpectogramDataset, preprocess_film_clap_data,
                                                         preprocess_tau_sed_data)
    from .dataset.synthetic import SyntheticSedDataset
    from .models.spectogram_models import Cnn_AvgPooling
    cfg = cfgs.REF_NATIVE
    name = args.dataset_name.lower()
    if name == "synthetic":
        dataset = SyntheticSedDataset(n_train_crops=max(256, 4 * args.batch_size), crop=cfg.train_crop_size * 8,
                                      classes=cfg.classes_num)
        descriptor = cfg.cfg_descriptor
    else:
        if name == "tau":
            feats_dir, mean_std = preprocess_tau_sed_data(args.dataset_dir, fold_name="eval",
                                                          preprocess_mode=args.preprocess_mode,
                                                          force_preprocess=args.force_preprocess, cfg=cfg)
            descriptor = cfg.cfg_descriptor + "_C-doorslam"
        elif name == "filmclap":
            feats_dir, mean_std = preprocess_film_clap_data(args.dataset_dir, preprocessed_mode=args.preprocess_mode,
                                                            force_preprocess=args.force_preprocess, cfg=cfg)
            descriptor = cfg.cfg_descriptor + "_tm-0.33"
        else:
            raise ValueError(f"Only tau and filmclap datasets are supported, '{args.dataset_name}' given")
        dataset = SpectogramDataset(feats_dir, mean_std, augment_data=args.augment_data,
                                    balance_classes=args.balance_classes,
                                    val_descriptor=_val_descriptor(args.val_descriptor),
                                    preprocessed_mode=args.preprocess_mode, cfg=cfg, device=device)
    model = Cnn_AvgPooling(cfg.classes_num, model_config=[(32, 2), (64, 2), (128, 2), (128, 1)])
    model.set_precision(args.precision)
    if args.ckpt != "":
        checkpoint = torch.load(args.ckpt, map_location=device)
        model.load_state_dict(checkpoint["model"])
    criterion = WeightedBCE(recall_factor=args.recall_priority, multi_frame=True)
    return dataset, model, criterion, f"{args.preprocess_mode}-{descriptor}"


def get_waveform_dataset_and_model(args, device):
    """main.py:49-73."""
    from .dataset.waveform.waveform_configs import cfg_descriptor, time_margin
    from .dataset.waveform.waveform_dataset import WaveformDataset, synthetic_waveform_task
    from .models.waveform_models import M5
    name = args.dataset_name.lower()
    waveforms = None
    if name == "synthetic":
        items, waveforms = synthetic_waveform_task()
        val_descriptor = "val_"
    elif name == "tau":
        from .dataset.dataset_utils import get_tau_sed_paths_and_labels, tau_audio_and_meta_dirs
        audio_dir, meta_dir = tau_audio_and_meta_dirs(f"{args.dataset_dir}/Tau_sound_events_2019", fold_name="eval")
        items, val_descriptor = get_tau_sed_paths_and_labels(audio_dir, meta_dir), _val_descriptor(args.val_descriptor)
    elif name == "filmclap":
        from .dataset.dataset_utils import get_film_clap_paths_and_labels
        items = get_film_clap_paths_and_labels(os.path.join(args.dataset_dir, "FilmClap"), time_margin)
        val_descriptor = _val_descriptor(args.val_descriptor)
    else:
        raise ValueError(f"Only tau and filmclap datasets are supported, '{args.dataset_name}' given")
    dataset = WaveformDataset(items, augment_data=args.augment_data, balance_classes=args.balance_classes,
                              val_descriptor=val_descriptor, waveforms=waveforms, device=device)
    model = M5(1, precision=args.precision)
    if args.ckpt != "":
        model.load_state_dict(torch.load(args.ckpt, map_location=device)["model"])
    criterion = WeightedBCE(recall_factor=args.recall_priority, multi_frame=False)
    return dataset, model, criterion, cfg_descriptor


def get_dataset_and_model(args, device):
    """main.py:77-83."""
    feats = args.train_features.lower()
    if feats == "spectogram":
        return get_spectogram_dataset_model_and_criterion(args, device)
    if feats == "waveform":
        return get_waveform_dataset_and_model(args, device)
    raise ValueError(f"training features can be raw waveform or spectogram only, '{args.train_features}' given")


def make_loader(dataset, batch_size):
    from .dataset.spectogram.spectograms_dataset import DeviceBatchLoader, SpectogramDataset
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if isinstance(dataset, SpectogramDataset):
        return DeviceBatchLoader(dataset, batch_size, rank=rank, world_size=world)
    from .dataset.waveform.waveform_dataset import WaveformBatchLoader, WaveformDataset
    if isinstance(dataset, WaveformDataset):
        return WaveformBatchLoader(dataset, batch_size, rank=rank, world_size=world,
                                   device=dataset.device or torch.device("cuda"))
    if world > 1:        # map-style dataset under data parallel: rank-sharded index ranges (a plain DataLoader would
        from .train import ShardedBatchLoader      # hand every rank the same batches)
        return ShardedBatchLoader(dataset, batch_size, rank=rank, world_size=world)
    from torch.utils.data import DataLoader
    return DataLoader(dataset, batch_size=batch_size, num_workers=0)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: this build has no CPU training path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device(f"cuda:{local_rank}" if world > 1 else args.device)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        from .train import seed_all_ranks
        dist.init_process_group("nccl")          # RCCL
        # the dataset classes shuffle / split with the global host RNGs (like the reference's): every rank must draw the
        # same train/val split and the same start-index permutation before the index ranges are sharded by rank
        seed = seed_all_ranks(int(os.environ["SED_SEED"]) if "SED_SEED" in os.environ else None)
    dataset, model, criterion, cfg_descriptor = get_dataset_and_model(args, device)
    if world > 1:
        from .train import reseed_rank
        reseed_rank(seed, int(os.environ.get("RANK", "0")))       # augmentation draws differ per rank from here on (split / index table are already drawn)
    dataloader = make_loader(dataset, args.batch_size)
    model = model.to(device)
    model.model_description()
    train_name = f"{args.dataset_name}_cfg({cfg_descriptor}_b{args.batch_size}_lr{args.lr}_{args.train_tag}"
    if args.balance_classes:
        train_name += "_BC"
    if args.augment_data:
        train_name += "_AD"
    train(model, dataloader, criterion, num_steps=args.num_train_steps,
          outputs_dir=os.path.join(args.outputs_root, train_name), device=device, lr=args.lr,
          log_freq=args.log_freq)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
