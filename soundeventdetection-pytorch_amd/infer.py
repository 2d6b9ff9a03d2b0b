"""Inference CLI (a working counterpart of /root/reference/infer.py:9-38, whose checkpoint load is
commented out and whose imports no longer resolve): WAV file -> log-mel on the MI355X -> Cnn_AvgPooling
in eval mode -> per-frame probabilities, threshold decisions and onset times.

    python -m sed_amd.infer recording.wav --ckpt training_dir/.../iteration_5000.pth

Writes <outputs_dir>/<name>.npz (probabilities, decisions, onset_frames, onset_seconds) and prints
the onsets.  The features are z-scored with --mean_std (the pickle the preprocessing wrote) when given:
the reference's infer.py skips the normalisation the model was trained with."""
from __future__ import annotations

import argparse
import os
import pickle

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description="SED inference on MI355X")
    p.add_argument("audio_file", type=str)
    p.add_argument("--ckpt", type=str, required=True)
    p.add_argument("--outputs_dir", type=str, default="inference_outputs", help="Directory of your workspace.")
    p.add_argument("--device", default="cuda:0", type=str)
    p.add_argument("--mean_std", type=str, default="", help="features_mean_std pickle of the training set")
    p.add_argument("--threshold", type=float, default=0.5)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "f16x3", "bf16x3"])
    return p


def onset_frames(decisions):
    """Rising edges of a 0/1 frame sequence (frame 0 counts when already active)."""
    d = np.asarray(decisions).astype(np.int8).reshape(-1)
    return np.flatnonzero(np.diff(np.concatenate(([0], d))) == 1)


def infer_file(audio_file, ckpt, device="cuda:0", mean_std="", threshold=0.5, precision="bf16"):
    from .dataset.dataset_utils import read_multichannel_audio
    from .dataset.spectogram.preprocess import LogMelFrontEnd
    from .dataset.spectogram.spectogram_configs import REF_NATIVE as cfg
    from .models.spectogram_models import Cnn_AvgPooling
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: this build has no CPU inference path")
    dev = torch.device(device)
    model = Cnn_AvgPooling(cfg.classes_num, model_config=[(32, 2), (64, 2), (128, 2), (128, 1)]).to(dev)
    model.set_precision(precision)
    checkpoint = torch.load(ckpt, map_location=dev)
    model.load_state_dict(checkpoint["model"] if "model" in checkpoint else checkpoint)
    model.eval()
    mean = std = None
    if mean_std:
        with open(mean_std, "rb") as f:
            d = pickle.load(f)
        mean, std = d["mean"], d["std"]
    print("Preprocessing audio file..")
    audio = read_multichannel_audio(audio_path=audio_file, target_fs=cfg.working_sample_rate, cfg=cfg)
    fe = LogMelFrontEnd(cfg, device=dev, mean=mean, std=std)
    feats = fe(np.ascontiguousarray(audio.T))                     # (channels, 1, T, mel) = (batch, 1, T, mel)
    print("Inference..")
    with torch.no_grad():
        logits = model(feats)                                       # (1, T', classes)
    probs = torch.sigmoid(logits)[0].cpu().numpy()
    dec = probs > threshold
    onsets = [onset_frames(dec[:, k]) for k in range(dec.shape[1])]
    return {"probabilities": probs, "decisions": dec, "onset_frames": onsets,
            "frames_per_second": cfg.frames_per_second, "log_mel": feats[0, 0].cpu().numpy()}


def main(argv=None):
    args = build_parser().parse_args(argv)
    res = infer_file(args.audio_file, args.ckpt, args.device, args.mean_std, args.threshold, args.precision)
    os.makedirs(args.outputs_dir, exist_ok=True)
    name = os.path.splitext(os.path.basename(args.audio_file))[0]
    fps = res["frames_per_second"]
    np.savez(os.path.join(args.outputs_dir, name + ".npz"), probabilities=res["probabilities"],
             decisions=res["decisions"], onset_frames=np.concatenate(res["onset_frames"]) if res["onset_frames"] else [],
             onset_seconds=np.concatenate(res["onset_frames"]) / fps if res["onset_frames"] else [])
    for k, on in enumerate(res["onset_frames"]):
        print(f"class {k}: {len(on)} onsets at " + ", ".join(f"{f / fps:.2f}s" for f in on[:50]))


if __name__ == "__main__":
    main()
