"""Multimodal training driver: drop-in for the reference's ``main_both.py`` (script body :236-834) on the HIP path.

    python main_both.py --late_fusion=MM_RCA --reverse --image_model=transformer_B16 --text_model=distilbert \
        --dataset_folder_name=Train --dataset_folder_name_val=Val --opt=sgd ...
    python -m torch.distributed.run --nproc-per-node 8 main_both.py ...      # one process per GPU, RCCL

Kept from the reference: every CLI flag (options.py), the two-phase schedule (frozen backbones for ``--epochs``, then
everything trainable for ``--ft_epochs`` at lr/``--fraction_lr``, main_both.py:562-832), ``run_one_epoch`` /
``calculate_set_accuracy`` semantics, four accuracy passes per epoch, ReduceLROnPlateau('max', factor=0.4) stepped in
the fine-tuning phase only, best-validation checkpointing with the reference's file-name pattern, wandb keys (wandb is
optional here: it needs network).  Replaced: ``nn.DataParallel`` -> sharded sampler + overlapped RCCL gradient
all-reduce; ``torch.optim`` -> fused flat-arena optimizers (same update rules).
Only ``--late_fusion=MM_RCA`` is constructible, as in the reference (SURVEY.md header).
"""
from __future__ import annotations

import os
import sys
import time
from datetime import datetime
from pathlib import Path

import numpy as np
import torch

from . import distributed as D
from .CustomImageTextFolder import CustomImageTextFolder, SyntheticImageTextDataset
from .multimodal_model import MM_RCA
from .optim import FlatAdamW, FlatSGD
from .options import args_parser
from .training import (calculate_set_accuracy, get_class_weights_from_counts, mode_config_dict, run_one_epoch)

_num_classes = 4
BASE_PATH = os.getcwd() + os.sep


class Transforms:
    """PIL image -> normalised CHW float tensor at the backbone's input size: the reference's VALIDATION pipeline
    (main_both.py:433-440) = PadToMaintainAR (keep_aspect_ratio.py:18-53, axis quirk included, see preprocess.py) ->
    Resize(INTER_LINEAR: half-pixel centres, no antialiasing, uint8 result) -> ImageNet mean/std.  This is the per-sample
    DataLoader-worker form; ``preprocess.GpuImagePipeline`` is the same arithmetic as one GPU launch per batch.  Of the
    training augmentations (albumentations, :407-429) only the vertical / horizontal flips with
    probability ``prob_aug`` are applied."""
    MEAN = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    STD = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    def __init__(self, width, height, train=False, prob_aug=0.0):
        self.width, self.height, self.train, self.prob_aug = width, height, train, prob_aug

    def __call__(self, img):
        from .preprocess import plan_padding
        arr = np.asarray(img.convert("RGB"))
        h, w = arr.shape[:2]
        pt, pl, ph, pw = plan_padding(h, w, self.width / self.height)
        if (ph, pw) != (h, w):
            arr = np.pad(arr, ((pt, ph - h - pt), (pl, pw - w - pl), (0, 0)))
        t = torch.from_numpy(arr.copy()).permute(2, 0, 1).unsqueeze(0).float()
        t = torch.nn.functional.interpolate(t, size=(self.height, self.width), mode="bilinear", align_corners=False,
                                            antialias=False)[0]
        t = torch.floor(t + 0.5).clamp_(0, 255) / 255.0          # cv2 returns uint8 for a uint8 image
        if self.train:
            if np.random.rand() < self.prob_aug:
                t = t.flip(1)
            if np.random.rand() < self.prob_aug:
                t = t.flip(2)
        return (t - self.MEAN) / self.STD


class DecodeOnly:
    """--gpu_preprocess: the DataLoader workers only decode (PIL -> uint8 HWC); padding, resize, the albumentations-style
    augmentations and the normalisation run on the GPU, per batch (preprocess.GpuImagePipeline)."""

    def __call__(self, img):
        return torch.from_numpy(np.array(img.convert("RGB"), dtype=np.uint8))


def collate_decoded(batch):
    """default_collate, except that the decoded images (different sizes) are packed back to back into one uint8 tensor
    (preprocess.pack_images: the GPU pipeline's staging layout, one shared-memory segment per batch)."""
    from .preprocess import pack_images
    raws = [sample[0]["image"].pop("raw_image") for sample in batch]
    data, labels = torch.utils.data.default_collate(batch)
    data["image"]["raw_image"] = pack_images(raws)
    return data, labels


def get_class_weights(train_dataset_path):
    """main_both.py:61-78."""
    ds = CustomImageTextFolder(train_dataset_path)
    return get_class_weights_from_counts([len(ds.per_class[i]) for i in range(_num_classes)])


def save_model_weights(model, text_model_name, image_model_name, epoch_num, val_acc, hw_device, fine_tuning,
                       class_weights, opt, fusion, args):
    """main_both.py:201-226 (file-name pattern kept).  state_dict tensors are copied to the host instead of moving the
    module."""
    base = os.path.join("model_weights", text_model_name + "_" + image_model_name)
    Path(os.path.join(BASE_PATH, base)).mkdir(parents=True, exist_ok=True)
    name = text_model_name + "_" + image_model_name
    if fine_tuning:
        filename = "BEST_model_{}_FT_EPOCH_{}_LR_{}_Reg_{}_FractionLR_{}_OPT_{}_VAL_ACC_{:.5f}".format(
            name, epoch_num + 1, args.lr, args.reg, args.fraction_lr, opt, val_acc)
    else:
        filename = "BEST_model_{}_epoch_{}_LR_{}_Reg_{}_VAL_ACC_{:.5f}_".format(name, epoch_num + 1, args.lr, args.reg, val_acc)
    filename = filename + "_" + fusion + "_" + datetime.now().strftime("%Y_%m_%d_%H_%M_%S")
    full_path = os.path.join(BASE_PATH, base, filename) + ".pth"
    print("Saving weights to {}".format(full_path))
    torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, full_path)
    return full_path


def count_parameters(model):
    return sum(p.numel() for p in model.parameters())


class _NoWandb:
    def init(self, **kw):
        return self

    def watch(self, *a, **kw):
        pass

    def log(self, d):
        print("[log]", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()})


def _wandb():
    try:
        import wandb
        return wandb
    except Exception:
        return _NoWandb()


def main(argv=None):
    args = args_parser(argv)
    rank, local, world = D.init_from_env()
    is_main = rank == 0
    if not torch.cuda.is_available():
        print("GPU not available!!!!  The MM-RCA HIP path has no CPU fallback.")
        sys.exit(1)
    if args.dataset_folder_name == "" and not args.synthetic:
        print("Please provide dataset path")
        sys.exit(1)
    if args.seed is not None:
        torch.manual_seed(args.seed); np.random.seed(args.seed)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    from .conv_engine import CONV_MODELS
    from . import spec as S
    if args.image_model not in S.VISION_SPECS and args.image_model not in CONV_MODELS:
        # the reference ignores --image_model and always builds EfficientNetV2-M (main_both.py:259, multimodal_model.py:188)
        print("Image model {!r}: using EffNetv2-Medium, as the reference does for every value of this flag".format(args.image_model))
        args.image_model = "EffNetv2-Medium"
    print("Text Model: {}".format(args.text_model))
    print("Image Model: {}".format(args.image_model))
    _batch_size, _batch_size_FT = args.batch_size, args.batch_size_FT
    if args.late_fusion != "MM_RCA":
        print("Wrong late fusion strategy: ", args.late_fusion)       # main_both.py:341-343
        sys.exit(1)
    global_model = MM_RCA(_num_classes, args.model_dropout, args.image_text_dropout, args.image_prob_dropout,
                          args.num_neurons_FC, args.text_model, _batch_size, args.reverse, args.features_only,
                          args.cross_attention_only, image_model_name=args.image_model,
                          dtype={"bf16": torch.bfloat16, "fp32": torch.float32}.get(args.dtype, args.dtype), device=device, image_size=args.image_size)
    print("Num total parameters of the model: {}".format(count_parameters(global_model)))
    wandb = _wandb() if is_main else _NoWandb()
    wandb.init(project="Garbage Classification Both - MI355X", config=dict(args.__dict__))
    WIDTH, HEIGHT = global_model.get_image_size()
    _tokenizer = global_model.get_tokenizer()
    _max_len = args.tokens_max_len or global_model.get_max_token_size()

    gpu_pre = False
    if args.synthetic:
        train_data = SyntheticImageTextDataset(args.synthetic, WIDTH, _max_len)
        val_data = SyntheticImageTextDataset(max(args.synthetic // 4, _batch_size), WIDTH, _max_len, seed_images=99, seed_text=77)
        class_weights = get_class_weights_from_counts([len(c) for c in train_data.per_class])
    else:
        train_path = os.path.join(BASE_PATH, args.dataset_folder_name)
        val_path = os.path.join(BASE_PATH, args.dataset_folder_name_val)
        class_weights = get_class_weights(train_path)
        gpu_pre = bool(args.gpu_preprocess)
        train_data = CustomImageTextFolder(root=train_path, transform=DecodeOnly() if gpu_pre else Transforms(WIDTH, HEIGHT, True, args.prob_aug),
                                           tokens_max_len=_max_len, tokenizer_text=_tokenizer, extended_desc=args.extended_desc_train)
        val_data = CustomImageTextFolder(root=val_path, transform=DecodeOnly() if gpu_pre else Transforms(WIDTH, HEIGHT), tokens_max_len=_max_len,
                                         tokenizer_text=_tokenizer, extended_desc=args.extended_desc_val)
    print("Class weights: {}".format(class_weights))

    # one shuffle seed for all ranks (random when --seed is not given, like the reference's unseeded loader); dropout streams
    # differ per rank so that replicas do not apply identical masks to their different samples
    shuffle_seed = D.broadcast_seed(args.seed, device if world > 1 and torch.distributed.get_backend() == "nccl" else "cpu")
    global_model._drop_seed += rank * 1000003
    print("Per-process batch size: {} (global batch = {} x {} ranks; lr unchanged)".format(_batch_size, _batch_size, world))
    if args.balanced_sampler:
        print("Using balanced sampler for training and validation sets")                  # main_both.py:478-479
    for flag, default in (("use_synonyms", False), ("prob_aug_text", 0.6)):
        if getattr(args, flag, default) != default:
            print("WARNING: --{} is accepted for CLI compatibility but not implemented on this path (SURVEY.md section 2: out of scope)".format(flag))

    # GPU input path: one pipeline object (two pinned + two device staging slots) for all loaders;
    # the training passes draw albumentations' parameters per image on the host, the evaluation passes draw none
    image_pipeline, aug_params = None, None
    if gpu_pre:
        from .preprocess import GpuImagePipeline, sample_train_params
        image_pipeline = GpuImagePipeline(HEIGHT, WIDTH, max_batch=max(_batch_size, _batch_size_FT), max_pixels=640 * 480, device=device)
        aug_rng = np.random.default_rng(None if args.seed is None else int(args.seed) + 7919 * (rank + 1))
        aug_params = lambda n: sample_train_params(aug_rng, n, args.prob_aug)      # noqa: E731
    else:
        print("CPU image transforms (--gpu_preprocess=false): of the training augmentations only the flips are applied")

    def loader(ds, bs, shuffle):
        if args.balanced_sampler:     # class-balanced draws with replacement, training AND validation loaders (main_both.py:481-526)
            sampler = D.BalancedShardedSampler(ds.targets, rank, world, seed=shuffle_seed + (0 if shuffle else 104729))
        else:
            sampler = D.ShardedSampler(len(ds), rank, world, shuffle=shuffle, seed=shuffle_seed)
        # GPU input path: workers come from a fork SERVER (a fresh interpreter without the GPU runtime), not from a fork of this
        # process -- every fork of a process with registered host memory makes the kernel driver evict and restore its GPU
        # queues; 16 workers cost 25-34 s of stalled GPU per loader start on the MI355X box (tools/input_bench.py)
        nw = D.loader_workers(args.num_workers)        # (capped by this rank's CPU slice, distributed.bind_rank_to_cpus)
        ctx = "forkserver" if (gpu_pre and nw > 0) else None
        return torch.utils.data.DataLoader(ds, batch_size=bs, sampler=sampler, num_workers=nw, pin_memory=True,
                                           collate_fn=collate_decoded if gpu_pre else None, multiprocessing_context=ctx,
                                           persistent_workers=bool(ctx)), sampler

    # The fine-tuning loaders are built when that phase starts, after the first phase's loaders are shut down: with persistent
    # (fork-server) workers all four at once would keep up to 4 x num_workers processes alive, the idle ones competing with the
    # decode workers that feed the GPU input path.
    (dl_train, s_train), (dl_val, _) = loader(train_data, _batch_size, True), loader(val_data, _batch_size, False)

    if args.opt == "adamw":
        optimizer = FlatAdamW(global_model, lr=args.lr, weight_decay=args.reg)
    elif args.opt == "sgd":
        optimizer = FlatSGD(global_model, lr=args.lr, weight_decay=args.reg)
    else:
        print("Invalid optimizer!")                                     # main_both.py:550-552
        sys.exit(1)
    sync = D.GradSync(global_model.engine.arena.g, world) if world > 1 else None
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'max', factor=0.4)
    state = dict(max_val=0.0, max_img=0.0, max_txt=0.0, best_epoch=0)
    graphs = {}                                # --hip_graph: the captured train steps, one per batch shape, kept across epochs

    def one_phase(n_epochs, dl_tr, sampler, dl_v, bs, acc_steps, fine_tuning):
        for epoch in range(n_epochs):
            sampler.set_epoch(epoch + (1000 if fine_tuning else 0))
            global_model.train()
            st = time.time()
            # auto: the launch-bound corner only -- one GPU, a batch of at most 16 images of at most 224 x 224 (configs[0]: 6.4 vs 18 ms per
            # step).  Measured at the reference's own launch shape (EfficientNetV2-M @ 480, B = 16) the step is GPU-bound: graph = eager
            # (469 vs 471 samples/s), and the eager step keeps the packed caption layout
            use_graph = args.hip_graph == "on" or (args.hip_graph == "auto" and bs <= 16 and world == 1 and bs * WIDTH * HEIGHT <= 16 * 224 * 224)
            _, losses = run_one_epoch(epoch, global_model, dl_tr, len(sampler), device, bs, optimizer, class_weights,
                                      args.balance_weights, acc_steps, args.label_smoothing, grad_sync=sync, verbose=is_main,
                                      image_pipeline=image_pipeline, aug_params=aug_params, hip_graph=(graphs if use_graph else None))
            elapsed = time.time() - st
            train_loss_avg = float(np.average([float(l) for l in losses])) if losses else float("nan")
            global_model.eval()
            ar = D.all_reduce_counts if world > 1 else None
            # eval_mode False: the TRAINING-branch modality dropout still fires in these two passes (main_both.py:594-619)
            train_acc, _ = calculate_set_accuracy(global_model, dl_tr, len(sampler), device, bs, mode_config_dict['both'], False, is_main, ar,
                                                  n_real=sampler.num_real, image_pipeline=image_pipeline, aug_params=aug_params)   # the train set is scored through its own (augmenting) pipeline, as in the reference
            val_acc, val_report = calculate_set_accuracy(global_model, dl_v, len(dl_v.sampler), device, bs, mode_config_dict['both'], False, is_main, ar,
                                                         n_real=dl_v.sampler.num_real, image_pipeline=image_pipeline)
            if val_acc > state["max_val"]:
                if is_main:
                    save_model_weights(global_model, args.text_model, args.image_model, epoch, val_acc, device, fine_tuning,
                                       args.balance_weights, args.opt, args.late_fusion, args)
                state["max_val"], state["best_epoch"] = val_acc, epoch
            img_only, _ = calculate_set_accuracy(global_model, dl_v, len(dl_v.sampler), device, bs, mode_config_dict['image_only'], True, is_main, ar,
                                                 n_real=dl_v.sampler.num_real, image_pipeline=image_pipeline)
            txt_only, _ = calculate_set_accuracy(global_model, dl_v, len(dl_v.sampler), device, bs, mode_config_dict['text_only'], True, is_main, ar,
                                                 n_real=dl_v.sampler.num_real, image_pipeline=image_pipeline)
            state["max_img"], state["max_txt"] = max(state["max_img"], img_only), max(state["max_txt"], txt_only)
            if fine_tuning:
                scheduler.step(val_acc)                                 # main_both.py:769
            log = {'epoch': epoch, 'epoch_time_seconds': elapsed, 'train_loss_avg': train_loss_avg,
                   'train_accuracy_history': train_acc, 'val_accuracy_history': val_acc,
                   'val_accuracy_text_only_history': txt_only, 'val_accuracy_image_only_history': img_only,
                   'max_val_acc': state["max_val"], 'max_img_only_val_acc': state["max_img"], 'max_txt_only_val_acc': state["max_txt"],
                   'samples_per_second': len(sampler) * world / max(elapsed, 1e-9)}
            for c in ("black", "blue", "green", "ttr"):
                log[c + '_val_precision'] = val_report.get(c, {}).get("precision", float("nan"))
            wandb.log(log)

    one_phase(args.epochs, dl_train, s_train, dl_val, _batch_size, args.acc_steps, False)
    print("Starting Fine tuning!!")
    if args.tl is True:
        for p in global_model.parameters():                             # main_both.py:690-697
            p.requires_grad = True
        for group in optimizer.param_groups:                            # :700-701
            group['lr'] = args.lr / args.fraction_lr
        for dl in (dl_train, dl_val):           # stop the first phase's persistent workers
            it = getattr(dl, "_iterator", None)
            if it is not None and hasattr(it, "_shutdown_workers"):
                it._shutdown_workers()
            dl._iterator = None
        del dl_train, dl_val
        (dl_train_ft, s_train_ft), (dl_val_ft, _) = loader(train_data, _batch_size_FT, True), loader(val_data, _batch_size_FT, False)
        one_phase(args.ft_epochs, dl_train_ft, s_train_ft, dl_val_ft, _batch_size_FT, args.acc_steps_FT, True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
