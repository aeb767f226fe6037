"""Drop-in for the reference's ``q_former_training.py``: same CLI (options.args_parser:
--dataset_folder_name, --dataset_folder_name_val, --batch_size, --epochs), same folder convention, same loop, on the HIP
engine of q_former.py.

    python -m garbage_classification_rca_amd.q_former_training --dataset_folder_name Train --dataset_folder_name_val Val \\
        --batch_size 64 --epochs 5 --blip2_checkpoint blip2-opt-2.7b.safetensors

Differences from the reference script, all forced by what is (not) on the path to the loss -- see q_former.py's docstring:
  * the prompt text (:74-81) feeds only the OPT language model, whose output the loss never reads: no tokenizer runs and
    the batch carries ``pixel_values`` and ``labels`` only;
  * the image side of ``AutoProcessor`` (:82; BlipImageProcessor with blip2-opt-2.7b's preprocessor_config: RGB, bicubic
    resize to 224x224, 1/255, CLIP mean / std) is restated in ``Blip2ImageTransform``;
  * ``save_checkpoint`` (:33-47) writes the classifier file under the reference's name; the 15 GB ``BLIP2_Q_FORMER_*.pth``
    would hold the unchanged frozen weights (no LoRA factor on the path ever gets a gradient) and is not rewritten;
  * wandb / tqdm / torchmetrics are not installed: accuracy is computed directly, progress goes to stdout.
"""
from __future__ import annotations

import glob
import os
import re
import sys
from typing import Dict, List

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import q_former as QF
from .options import args_parser

strings = ['Blue', 'Green', 'Black', 'Yellow']                    # q_former_training.py:258-263
cls_dict = {"Blue": 0, "Green": 1, "Black": 2, "Yellow": 3}

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def remove_numbers(input_string: str) -> str:                     # :58-59
    return re.sub(r'\d+', '', input_string)


class Blip2ImageTransform:
    """BlipImageProcessor as configured for Salesforce/blip2-opt-2.7b: convert to RGB, resize to size x size (PIL bicubic),
    rescale by 1/255, normalise with the CLIP statistics.  Returns fp32 [3, size, size]."""

    def __init__(self, size: int = 224):
        self.size = size
        self.mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
        self.std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)

    def __call__(self, img) -> torch.Tensor:
        from PIL import Image
        img = img.convert("RGB").resize((self.size, self.size), resample=Image.BICUBIC)
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1.0 / 255.0)
        return torch.from_numpy((a - self.mean) / self.std)


class ImageCaptioningDataset(Dataset):
    """:62-92.  Item: {'pixel_values': [1,3,H,W], 'labels': [1]}; label = the parent folder's name, TTR -> Yellow (:86-89).
    ``item_text`` (the file name without digits / extension, :72) is kept as an attribute of the item for inspection only."""

    def __init__(self, paths: List[str], processor=None, image_size: int = 224):
        self.dataset = paths
        self.processor = processor or Blip2ImageTransform(image_size)

    def __len__(self):
        return len(self.dataset)

    def item_text(self, idx: int) -> str:
        return remove_numbers(self.dataset[idx].split("/")[-1])[:-4].replace("_", " ").rstrip().lstrip()

    def __getitem__(self, idx):
        from PIL import Image
        item_path = self.dataset[idx]
        pixel_values = self.processor(Image.open(item_path)).unsqueeze(0)
        label = item_path.split('/')[-2]
        if label == "TTR":
            label = "Yellow"
        return {"pixel_values": pixel_values, "labels": (torch.ones(1) * cls_dict[label]).long()}      # gen_inputs, :49-53


def collate_fn(batch) -> Dict[str, torch.Tensor]:
    """:94-122 for the keys on the path: stack, then drop the per-item leading axis of pixel_values."""
    return {"pixel_values": torch.stack([b["pixel_values"] for b in batch]).squeeze(1),
            "labels": torch.stack([b["labels"] for b in batch])}


def load_blip2_state(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location="cpu")
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def save_checkpoint(engine: QF.Blip2QFormerEngine, epoch: int, acc: float, out_dir: str = ".") -> str:
    """:33-47: Classifier_epoch_<e>_acc_<a>.pth (keys classifier.weight / classifier.bias, as MultimodalClassifier saves)."""
    fn = os.path.join(out_dir, "Classifier_epoch_" + str(epoch) + "_acc_" + str(acc) + ".pth")
    print("Saving weights to {}".format(fn))
    torch.save(engine.classifier_state_dict(), fn)
    return fn


def main(argv=None, spec: QF.Blip2Spec = QF.BLIP2_OPT_2_7B, out_dir: str = "."):
    args = args_parser(argv)
    device = torch.device("cuda:0")                                                        # :201
    # --dtype (options.py): bf16x3f by default = the reference's fp32 arithmetic (q_former_training.py:279-304 runs fp32) to ~1e-5 on the
    # bf16 matrix cores; bf16 is the opt-in fast mode (logits ~2e-2 off at full depth)
    dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}.get(args.dtype, "bf16x3f")
    engine = QF.Blip2QFormerEngine(spec, dtype=dtype, device=device)
    if args.blip2_checkpoint:
        engine.init_parameters(seed=0)          # the classifier's nn.Linear default init; the frozen part is overwritten below
        engine.load_state_dict(load_blip2_state(args.blip2_checkpoint))
    else:
        print("WARNING: no --blip2_checkpoint given: the frozen BLIP-2 weights are RANDOM (the reference downloads "
              "Salesforce/blip2-opt-2.7b, :203-205; there is no network here)", file=sys.stderr)
        engine.init_parameters(seed=0)
    if args.classifier_weights:
        engine.load_state_dict({}, torch.load(args.classifier_weights, map_location="cpu"), strict=False)
    ims = sorted(glob.glob(args.dataset_folder_name + "/*/*"))                             # :208
    ims_val = sorted(glob.glob(args.dataset_folder_name_val + "/*/*"))                     # :252
    if not ims:
        raise FileNotFoundError(f"no images under {args.dataset_folder_name}/*/*")
    workers = min(32, args.num_workers)                                                    # :212
    mk = dict(batch_size=args.batch_size, num_workers=workers, collate_fn=collate_fn, pin_memory=True,
              multiprocessing_context="forkserver" if workers > 0 else None, persistent_workers=workers > 0)
    loader_train = DataLoader(ImageCaptioningDataset(ims, image_size=spec.image_size), shuffle=True, **mk)          # :214-215
    loader_val = DataLoader(ImageCaptioningDataset(ims_val, image_size=spec.image_size), shuffle=False, **mk) if ims_val else None
    optimizer = QF.ClassifierAdamW(engine)                                                 # :243-244
    max_val_accuracy, best_epoch, history = 0.0, 0, []
    for epoch in range(args.epochs):                                                       # :270
        avg_loss = QF.run_one_epoch(engine, optimizer, loader_train, device)
        print("loss", avg_loss)
        train_acc = QF.calculate_acc(engine, loader_train, device)                         # :313
        val_accuracy = QF.calculate_acc(engine, loader_val, device) if loader_val is not None else train_acc
        print(f"Epoch {epoch + 1}/{args.epochs}: train acc {train_acc:.4f}, validation acc {val_accuracy:.4f}")
        if val_accuracy > max_val_accuracy:                                                # :316-320
            print("Best model obtained based on Val Acc. Saving it!")
            save_checkpoint(engine, epoch, val_accuracy, out_dir)
            max_val_accuracy, best_epoch = val_accuracy, epoch
        else:
            print("Not saving model on epoch {}, best Val Acc so far on epoch {}: {:.3f}".format(epoch, best_epoch, max_val_accuracy))
        history.append({"train_loss_avg": avg_loss, "train_accuracy_history": train_acc, "val_accuracy_history": val_accuracy,
                        "max_val_acc_percentage": max_val_accuracy * 100})                 # the wandb.log payload, :325-328
    return history


if __name__ == "__main__":
    main()
