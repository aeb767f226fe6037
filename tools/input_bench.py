"""End-to-end input path on a real folder: JPEG files -> DataLoader workers -> images in HBM,
with the reference-style per-sample CPU transforms vs decode-only workers + the GPU pipeline (all augmentations), and one
real training epoch of MM-RCA (ViT-B/16 + DistilBERT, bf16) fed by the GPU pipeline.

    python tools/input_bench.py [--n 4096] [--workers 16] [--batch 256] [--skip_epoch]
Prints one JSON line."""
import argparse, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image

from garbage_classification_rca_amd.CustomImageTextFolder import CustomImageTextFolder
from garbage_classification_rca_amd.main_both import DecodeOnly, Transforms, collate_decoded
from garbage_classification_rca_amd.preprocess import GpuImagePipeline, sample_train_params


def make_folder(root, n, rng):
    names = ["chip bag", "pizza box", "banana peel", "aa batteries", "glass jar", "paper cup"]
    base = (rng.random((64, 64, 3)) * 255).astype(np.uint8)
    for i in range(n):
        c = ["Black", "Blue", "Green", "TTR"][i % 4]
        d = os.path.join(root, c)
        os.makedirs(d, exist_ok=True)
        h, w = int(rng.integers(300, 520)), int(rng.integers(300, 520))
        img = Image.fromarray(base).resize((w, h), Image.BILINEAR)
        arr = np.asarray(img).copy()
        arr[:: 7, :, i % 3] = (i * 37) % 256
        Image.fromarray(arr).save(os.path.join(d, f"{names[i % len(names)]}_{i}.jpg"), quality=90)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--skip_epoch", action="store_true")
    ap.add_argument("--pin", type=int, default=1, help="DataLoader pin_memory for the GPU path")
    ap.add_argument("--ctx", default="forkserver", help="multiprocessing context of the GPU path's workers (fork | forkserver | spawn)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tmp = tempfile.mkdtemp(prefix="mmrca_inputs_")
    res = {"pin": a.pin, "ctx": a.ctx, "n_images": a.n, "workers": a.workers, "batch": a.batch, "image_sizes": "300-520 px JPEG q90"}
    try:
        t0 = time.time()
        make_folder(tmp, a.n, np.random.default_rng(0))
        res["make_folder_s"] = round(time.time() - t0, 1)
        from garbage_classification_rca_amd.multimodal_model import MM_RCA
        from garbage_classification_rca_amd.optim import FlatAdamW
        from garbage_classification_rca_amd.training import run_one_epoch
        model = MM_RCA(4, 0.3, 0.0, 0.0, 256, "distilbert", a.batch, True, False, False, image_model_name="transformer_B16",
                       dtype=torch.bfloat16, device=dev)
        tok = model.get_tokenizer()                      # HF files when cached, else the offline hashing tokenizer

        def run(ds, collate, consume):
            dl = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, num_workers=a.workers, collate_fn=collate,
                                             pin_memory=(collate is None or bool(a.pin)), persistent_workers=False,
                                             multiprocessing_context=None if collate is None else a.ctx)
            t = time.time()
            stamps = []
            for data, _ in dl:
                x = consume(data["image"]["raw_image"])
                torch.cuda.synchronize()
                stamps.append((time.time(), x.shape[0]))
            run.startup = round(stamps[0][0] - t, 2)        # worker start-up + first batch
            # steady state = the second half of the epoch: by then the batches the workers prefetched while starting up are used up
            half = len(stamps) // 2
            return sum(b for _, b in stamps[half:]) / (stamps[-1][0] - stamps[half - 1][0])

        cpu_ds = CustomImageTextFolder(tmp, tokens_max_len=24, tokenizer_text=tok, transform=Transforms(224, 224, True, 0.5))
        res["cpu_transforms_images_per_s"] = round(run(cpu_ds, None, lambda t: t.to(dev, non_blocking=True)), 1)
        gpu_ds = CustomImageTextFolder(tmp, tokens_max_len=24, tokenizer_text=tok, transform=DecodeOnly())
        pipe = GpuImagePipeline(224, 224, max_batch=a.batch, max_pixels=520 * 520, device=dev)
        rng = np.random.default_rng(1)
        res["cpu_startup_s"] = run.startup
        res["gpu_pipeline_images_per_s"] = round(run(gpu_ds, collate_decoded, lambda raws: pipe(raws, aug=sample_train_params(rng, len(raws["shapes"]), 0.5))), 1)
        res["gpu_startup_s"] = run.startup
        # the GPU stages alone (images already decoded in host memory)
        raws = [gpu_ds[i][0]["image"]["raw_image"] for i in range(a.batch)]
        params = sample_train_params(rng, a.batch, 0.5)
        for _ in range(3):
            pipe(raws, aug=params)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            pipe(raws, aug=params)
        torch.cuda.synchronize()
        res["gpu_stages_only_images_per_s"] = round(10 * a.batch / (time.time() - t), 1)
        if not a.skip_epoch:
            opt = FlatAdamW(model, lr=1e-5, weight_decay=0.01)
            for p_ in model.parameters():                  # the fine-tuning phase: both encoders train (the benchmarked step)
                p_.requires_grad = True
            dl = torch.utils.data.DataLoader(gpu_ds, batch_size=a.batch, shuffle=True, num_workers=a.workers, collate_fn=collate_decoded,
                                             pin_memory=bool(a.pin), multiprocessing_context=a.ctx, persistent_workers=True)
            model.train()
            import contextlib, io
            for ep in range(2):                          # epoch 0 pays the worker start-up; epoch 1 is the steady state (persistent workers)
                t = time.time()
                with contextlib.redirect_stdout(io.StringIO()):
                    run_one_epoch(ep, model, dl, len(gpu_ds), dev, a.batch, opt, [1, 1, 1, 1], False, 0, 0.0, verbose=False, image_pipeline=pipe,
                                  aug_params=lambda n: sample_train_params(rng, n, 0.5))
                torch.cuda.synchronize()
                res["train_epoch%d_images_per_s" % ep] = round(len(gpu_ds) / (time.time() - t), 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
