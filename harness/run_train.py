#!/usr/bin/env python3
"""Counterpart of the reference's ``Signal_vs_Noise/run_train.py`` + ``src/train.py`` on the MI355X path.

Same flags (``run_train.py:9-25``), same construction sequence (``src/train.py:227-277``: encoder -> fnmatch target
search -> ``LoraConfig(use_dora=True)`` -> ``get_peft_model`` -> ``requires_grad = 'lora' in name`` ->
``two_channel_ligo_binary_classifier`` -> ``BCEWithLogitsLoss`` + ``AdamW(lr, betas=(0.9, 0.999), eps=1e-8)``), same
loop (``:139-211``: train epoch, validation loss + ROC AUC, best-by-val-loss checkpoint, early stopping with
patience 15) and the same artefacts (``<models-path>/[best_]lora_weights_<r>_<alpha>/adapter_config.json +
adapter_model.safetensors``, ``[best_]dense_layers_<r>_<alpha>.pth``).

Differences that come with the hardware path:
  * log-mel features are computed on the GPU per batch (``ops.logmel``) from the 16 kHz waveforms, not per item in
    DataLoader workers (``src/dataset.py:15-26``);
  * ``--data-path`` is a HuggingFace ``datasets`` directory with the reference's columns (``h1_timeseries``,
    ``l1_timeseries``, ``labels``, ``injection_snr``; ``utils/preprocess.py:111-132``) or, with ``--synthetic N``,
    N seeded noise segments of which half carry a chirp (there is no network for real data);
  * pretrained ``openai/whisper-*`` weights cannot be downloaded here: ``--encoder-weights`` takes a HF encoder
    ``state_dict`` (.pth / .safetensors), otherwise seeded random weights;
  * scalars go to ``<log-dir>/train_log.jsonl`` (TensorBoard is not installed);
  * under ``torchrun`` every rank trains on its shard of each epoch and the trainable gradients are all-reduced
    in ONE flat bucket (RCCL).
``--method DoRA`` (the reference default) and ``--method LoRA`` have a HIP backward; ``full_finetune`` raises.
"""
import argparse
import fnmatch
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


class EarlyStopper:   # src/train.py:27-43
    def __init__(self, patience=1, min_delta=0.0):
        self.patience, self.min_delta, self.counter, self.min_validation_loss = patience, min_delta, 0, float("inf")

    def early_stop(self, validation_loss):
        if validation_loss < self.min_validation_loss:
            self.min_validation_loss, self.counter = validation_loss, 0
        elif validation_loss > (self.min_validation_loss + self.min_delta):
            self.counter += 1
            if self.counter >= self.patience:
                return True
        return False


def synthetic_dataset(n, seed):
    """n two-detector 1 s segments at 16 kHz, unit-variance noise; odd indices carry the same chirp in both."""
    from gw_whisper_amd import synth
    h1 = synth.strain_segments(n, seed=seed)
    l1 = synth.strain_segments(n, seed=seed + 1)
    t = np.arange(16000, dtype=np.float32) / 16000.0
    labels = np.zeros(n, np.float32)
    rng = np.random.default_rng(seed + 2)
    for i in range(1, n, 2):
        f0, tc = 40.0 + 60.0 * rng.random(), 0.45 + 0.3 * rng.random()
        s = 4.0 * np.sin(2 * np.pi * (f0 + 220.0 * t) * t) * np.exp(-((t - tc) / 0.12) ** 2)
        h1[i] += s.astype(np.float32)
        l1[i] += s.astype(np.float32)
        labels[i] = 1.0
    return h1, l1, labels, np.where(labels > 0, 12.0, 0.0).astype(np.float32)


def load_arrays(args):
    if args.synthetic:
        return synthetic_dataset(args.synthetic, args.seed)
    from datasets import concatenate_datasets, load_from_disk
    path = args.data_path
    chunks = sorted(p for p in os.listdir(path) if p.startswith("chunk")) if os.path.isdir(path) else []
    ds = concatenate_datasets([load_from_disk(os.path.join(path, c)) for c in chunks]) if chunks else load_from_disk(path)
    cols = ds.with_format("numpy")
    return (np.asarray(cols["h1_timeseries"], np.float32), np.asarray(cols["l1_timeseries"], np.float32),
            np.asarray(cols["labels"], np.float32), np.asarray(cols["injection_snr"], np.float32))


def main(args):
    from gw_whisper_amd import dist as gdist, ops, synth
    from gw_whisper_amd.encoder import WhisperConfig, WhisperEncoder
    from gw_whisper_amd.models import two_channel_ligo_binary_classifier
    from gw_whisper_amd.peft import LoraConfig, get_peft_model
    rank, world, local = gdist.init()
    assert torch.cuda.is_available(), "run_train.py needs an MI355X (gw_whisper_amd has no CPU path)"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)

    h1, l1, labels, snr = load_arrays(args)
    n = len(labels)
    perm = np.random.default_rng(args.seed).permutation(n)           # train_test_split(test_size, seed, shuffle=True)
    n_val = int(round(n * args.test_size))
    val_idx, train_idx = perm[:n_val], perm[n_val:]

    d, L, H, F = synth.ENCODER_SIZES[args.encoder]
    encoder = WhisperEncoder(WhisperConfig(d, L, H, F), precision="bf16")
    if args.encoder_weights:
        if args.encoder_weights.endswith(".safetensors"):
            from safetensors.torch import load_file
            encoder.load_state_dict(load_file(args.encoder_weights))
        else:
            encoder.load_state_dict(torch.load(args.encoder_weights, map_location="cpu"))
    else:
        sd = synth.encoder_state_dict(d, L, H, F, seed=args.seed)
        encoder.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    module_names = [name for name, _ in encoder.named_modules()]
    patterns = ["layers.*.self_attn.q_proj", "layers.*.self_attn.k_proj", "layers.*.self_attn.v_proj",
                "layers.*.self_attn.o_proj"]                          # src/train.py:232 (o_proj matches nothing in HF Whisper)
    matched = [m for p in patterns for m in fnmatch.filter(module_names, p)]
    if args.method not in ("DoRA", "LoRA"):
        # src/train.py:244-250: full_finetune trains every base weight -- that needs weight-gradient GEMMs for all 40
        # dense panels and the stem, which this hot path (frozen base) does not build
        raise NotImplementedError("--method DoRA and LoRA have a HIP backward; full_finetune is out of scope (DESIGN.md section 6)")
    if args.load_model_path:
        # resume (src/train.py:44-60): the saved adapter is loaded onto the BARE encoder -- PeftModel.from_pretrained
        # wraps the nn.Linear targets itself -- and stays trainable
        from gw_whisper_amd.peft import PeftModel
        adapter_dir = _resume_path(args.load_model_path, args.load_lora_weights)
        with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
            saved_dora = bool(json.load(f).get("use_dora", False))
        if saved_dora != (args.method == "DoRA"):
            raise ValueError(f"--method {args.method} but the adapter in {adapter_dir} was saved with use_dora={saved_dora}: "
                             "resume with the method it was trained with")
        peft = PeftModel.from_pretrained(encoder, adapter_dir, is_trainable=True).to(device)
    else:
        peft = get_peft_model(encoder, LoraConfig(use_dora=args.method == "DoRA", r=args.lora_rank,   # src/train.py:253, :263
                                                  lora_alpha=args.lora_alpha, target_modules=matched)).to(device)
    for name, p in peft.named_parameters():
        p.requires_grad = "lora" in name
    model = two_channel_ligo_binary_classifier(peft).to(device)
    if args.load_model_path:
        model.classifier.load_state_dict(torch.load(_resume_path(args.load_model_path, args.load_dense_weights),
                                                    map_location=device))
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.AdamW(params, lr=args.learning_rate, betas=(0.9, 0.999), eps=1e-08)
    bucket = gdist.FlatGradBucket(params)
    criterion = torch.nn.BCEWithLogitsLoss().to(device)
    tag = f"{args.lora_rank}_{args.lora_alpha}"
    os.makedirs(args.models_path, exist_ok=True)
    os.makedirs(args.log_dir, exist_ok=True)
    log = open(os.path.join(args.log_dir, "train_log.jsonl"), "a") if rank == 0 else None

    def features(idx):
        w = torch.from_numpy(np.concatenate((h1[idx], l1[idx]))).to(device)
        mel = ops.logmel(w)
        return mel[: len(idx)], mel[len(idx):]

    def evaluate():
        model.eval()
        tot, cnt, probs, ys = 0.0, 0, [], []
        with torch.no_grad():
            for i in range(0, len(val_idx), args.batch_size):
                idx = val_idx[i:i + args.batch_size]
                a, b = features(idx)
                out = model(a, b)
                y = torch.from_numpy(labels[idx]).view(-1, 1).to(device)
                tot += criterion(out, y).item() * len(idx)
                cnt += len(idx)
                probs.append(torch.sigmoid(out).flatten().cpu().numpy())
                ys.append(labels[idx])
        auc = float("nan")
        if cnt and len(set(np.concatenate(ys).tolist())) > 1:
            from sklearn.metrics import roc_auc_score
            auc = float(roc_auc_score(np.concatenate(ys), np.concatenate(probs)))
        return tot / max(cnt, 1), auc

    def save(prefix):
        if rank != 0:
            return
        model.encoder.save_pretrained(os.path.join(args.models_path, f"{prefix}lora_weights_{tag}"))
        torch.save(model.classifier.state_dict(), os.path.join(args.models_path, f"{prefix}dense_layers_{tag}.pth"))

    stopper, best = EarlyStopper(patience=15), float("inf")
    for epoch in range(args.num_epochs):
        model.train()
        order = np.random.default_rng(args.seed + 1 + epoch).permutation(train_idx)
        t0, run, nb, seen = time.time(), 0.0, 0, 0
        # every rank runs the SAME number of steps (one all-reduce each); the bucket takes the SAMPLE-weighted mean over
        # the ranks (a rank without a batch in the last step contributes zero samples), i.e. the gradient of the mean loss
        # over the whole global batch whatever the number of ranks
        for step in range(gdist.epoch_steps(len(order), world, args.batch_size)):
            sl = gdist.step_slice(len(order), step, rank, world, args.batch_size)
            bucket.zero()
            if sl is not None:
                idx = order[sl[0]:sl[1]]
                a, b = features(idx)
                y = torch.from_numpy(labels[idx]).view(-1, 1).to(device)
                loss = criterion(model(a, b), y)
                loss.backward()
                run += loss.item()
                nb += 1
                seen += len(idx)
            bucket.all_reduce_mean(world, n_local=0 if sl is None else sl[1] - sl[0])
            optimizer.step()
        train_loss = run / max(nb, 1)
        val_loss, val_auc = evaluate()
        val_loss = gdist.broadcast_scalar(val_loss, world, device)     # one decision for all ranks
        rec = {"epoch": epoch + 1, "train_loss": train_loss, "val_loss": val_loss, "val_auc": val_auc,
               "epoch_s": time.time() - t0, "segments_per_s": len(order) / max(time.time() - t0, 1e-9), "rank0_segments": seen}
        if rank == 0:
            print(f"Epoch {epoch + 1}/{args.num_epochs}, Train Loss: {train_loss:.4f}, Val Loss: {val_loss:.4f}, "
                  f"Val AUC: {val_auc:.4f}")
            log.write(json.dumps(rec) + "\n")
            log.flush()
        if val_loss < best:
            best = val_loss
            save("best_")
        if stopper.early_stop(val_loss):
            if rank == 0:
                print(f"Early stopping at epoch {epoch + 1}")
            break
    save("")
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _resume_path(base: str, name: str) -> str:
    """The reference builds resume paths by plain string concatenation, ``load_model_path + load_lora_weights``
    (Signal_vs_Noise/src/train.py:292): that form first, ``os.path.join`` (a base directory without the trailing
    separator) second; whichever exists."""
    for cand in (base + name, os.path.join(base, name)):
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(f"neither {base + name!r} nor {os.path.join(base, name)!r} exists")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Training pipeline for the LIGO binary classification model (MI355X path)")
    parser.add_argument("--data-path", type=str, default="Detection/data/Whisper_train_mass-8to100_resampled_train")
    parser.add_argument("--models-path", type=str, default="Detection/results/Two_detectors/models")
    parser.add_argument("--figures-path", type=str, default="Detection/results/Two_detectors/figures")
    parser.add_argument("--log-dir", type=str, default="Detection/results/Two_detectors/logs")
    parser.add_argument("--load_model_path", type=str, default=None)
    parser.add_argument("--load_lora_weights", type=str, default=None)
    parser.add_argument("--load_dense_weights", type=str, default=None)
    parser.add_argument("--batch-size", type=int, default=32)
    parser.add_argument("--num-epochs", type=int, default=1)
    parser.add_argument("--learning-rate", type=float, default=1e-4)
    parser.add_argument("--test-size", type=float, default=0.2)
    parser.add_argument("--encoder", type=str, default="tiny")
    parser.add_argument("--seed", type=int, default=42)
    parser.add_argument("--num-workers", type=int, default=12, help="accepted for compatibility; features are computed on the GPU")
    parser.add_argument("--method", type=str, default="DoRA")
    parser.add_argument("--lora-rank", type=int, default=8)
    parser.add_argument("--lora-alpha", type=int, default=32)
    parser.add_argument("--synthetic", type=int, default=0, help="use N seeded synthetic segments instead of --data-path")
    parser.add_argument("--encoder-weights", type=str, default=None, help="HF WhisperEncoder state_dict (.pth / .safetensors)")
    main(parser.parse_args())
