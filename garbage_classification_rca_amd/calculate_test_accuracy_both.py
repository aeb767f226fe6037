"""Test-split evaluator: the reference's ``calculate_test_accuracy_both.py`` on the HIP module.

    python calculate_test_accuracy_both.py --late_fusion=MM_RCA --reverse --image_model=transformer_B16 \
        --text_model=distilbert --model_path=<checkpoint.pth> --dataset_folder_name=<Test folder>

Kept: ``calculate_test_accuracy`` (:52-117: argmax, running accuracy, confusion matrix, sklearn classification report with
the class names Black/Blue/Green/TTR), the CSV report and confusion-matrix image file names (:119-141), seeds 42
(:151-157), batch 16, eval with both modalities.  Fixed: the reference passes 9 arguments to ``MM_RCA`` here (:162-171,
``cross_attention_only`` missing) and cannot construct the model; this script passes all ten.  The confusion matrix is
computed with numpy (torchmetrics / seaborn are optional).
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

from .CustomImageTextFolder import CustomImageTextFolder
from .main_both import DecodeOnly, Transforms, collate_decoded
from .training import stage_images, trim_caption_columns
from .multimodal_model import MM_RCA
from .options import args_parser
from .training import mode_config_dict

_num_classes = 4
BASE_PATH = os.getcwd() + os.sep
classes = ["Black", "Blue", "Green", "TTR"]


def confusion_matrix(labels, preds, n=_num_classes) -> np.ndarray:
    cm = np.zeros((n, n), dtype=np.int64)
    for t, p in zip(labels, preds):
        cm[int(t), int(p)] += 1
    return cm


def calculate_test_accuracy(model, data_loader, len_test_data, hw_device, batch_size, mode, eval_mode, verbose=True, image_pipeline=None):
    """Reference :52-117.  Returns (accuracy %, text report, report dict, confusion matrix)."""
    correct = 0
    n_batches = math.ceil(len_test_data / batch_size)
    all_preds, all_labels = [], []
    with torch.no_grad():
        for batch_idx, (data, labels) in enumerate(data_loader):
            texts = data['text']
            images = stage_images(data['image']['raw_image'], hw_device, image_pipeline)
            tok, msk = texts['tokens'], texts['attention_mask']
            if hasattr(model, "engine") and not msk.is_cuda:      # drop the padding columns no caption of the batch uses (training.py)
                tok, msk = trim_caption_columns(tok, msk)
            ids, mask = tok.to(hw_device), msk.to(hw_device)
            labels = labels.to(hw_device)
            outputs = model(_input_ids=ids, _attention_mask=mask, _images=images, eval=eval_mode,
                            remove_text=mode["remove_text"], remove_image=mode["remove_image"])
            pred = torch.max(outputs, 1)[1].view(-1)
            all_preds += pred.cpu().tolist()
            all_labels += labels.cpu().tolist()
            correct += torch.sum(torch.eq(pred, labels)).item()
            if verbose:
                print("Test batches {}/{} ".format(batch_idx, n_batches))
                print("Running test accuracy: {:.3f} %".format(100 * (correct / len_test_data)))
    test_acc = 100 * (correct / max(len_test_data, 1))
    cm = confusion_matrix(all_labels, all_preds)
    try:
        from sklearn.metrics import classification_report
        kw = dict(labels=list(range(_num_classes)), target_names=classes, zero_division=0)
        report = classification_report(all_labels, all_preds, **kw)
        report_dict = classification_report(all_labels, all_preds, output_dict=True, **kw)
    except Exception:
        report, report_dict = "", {}
    return test_acc, report, report_dict, cm


def generate_report_and_image(test_report_dict, test_accuracy, conf_matrix, mode, out_dir=None):
    """Reference :119-141 (same file names)."""
    out_dir = out_dir or BASE_PATH
    import pandas as pd
    csv_path = os.path.join(out_dir, "multimodal_model_report_test_set_acc_{:.2f}_{}.csv".format(test_accuracy, mode))
    pd.DataFrame.from_dict(test_report_dict).to_csv(csv_path, index=True)
    png_path = os.path.join(out_dir, 'conf_matrix_multimodal_model_test_set_acc_{:.2f}_{}.png'.format(test_accuracy, mode))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.rcParams.update({'font.size': 16})
        fig, ax = plt.subplots(figsize=(10, 5))
        im = ax.imshow(conf_matrix, cmap='viridis')
        ax.set_xticks(range(_num_classes)); ax.set_yticks(range(_num_classes))
        ax.set_xticklabels(classes); ax.set_yticklabels(classes)
        for i in range(_num_classes):
            for j in range(_num_classes):
                ax.text(j, i, str(int(conf_matrix[i, j])), ha="center", va="center", color="w")
        fig.colorbar(im)
        fig.savefig(png_path)
        plt.close(fig)
    except Exception as e:      # plotting is optional
        print(f"[mmrca] confusion-matrix image skipped ({type(e).__name__}: {e})")
        png_path = None
    return csv_path, png_path


def main(argv=None):
    args = args_parser(argv)
    if not torch.cuda.is_available():
        print("GPU not available!!!!  The MM-RCA HIP path has no CPU fallback.")
        sys.exit(1)
    device = torch.device("cuda:0")
    torch.manual_seed(42)
    np.random.seed(42)
    _batch_size = 16
    if args.late_fusion != "MM_RCA":
        print("Wrong late fusion strategy: ", args.late_fusion)
        sys.exit(1)
    from .conv_engine import CONV_MODELS
    from . import spec as S
    image_model = args.image_model if (args.image_model in S.VISION_SPECS or args.image_model in CONV_MODELS) else "EffNetv2-Medium"
    model = MM_RCA(_num_classes, args.model_dropout, args.image_text_dropout, args.image_prob_dropout, args.num_neurons_FC,
                   args.text_model, _batch_size, args.reverse, args.features_only, args.cross_attention_only,
                   image_model_name=image_model, dtype={"bf16": torch.bfloat16, "fp32": torch.float32}.get(args.dtype, args.dtype), device=device,
                   image_size=args.image_size)
    model.load_state_dict(torch.load(args.model_path, map_location=device))
    model.eval()
    WIDTH, HEIGHT = model.get_image_size()
    test_data = CustomImageTextFolder(root=args.dataset_folder_name, tokens_max_len=args.tokens_max_len or model.get_max_token_size(),
                                      tokenizer_text=model.get_tokenizer(),
                                      transform=DecodeOnly() if args.gpu_preprocess else Transforms(WIDTH, HEIGHT))
    print("Num of test samples: {}".format(len(test_data)))
    image_pipeline = None
    if args.gpu_preprocess:
        from .preprocess import GpuImagePipeline
        image_pipeline = GpuImagePipeline(HEIGHT, WIDTH, max_batch=_batch_size, max_pixels=640 * 480, device=device)
    loader = torch.utils.data.DataLoader(dataset=test_data, batch_size=_batch_size, shuffle=True, num_workers=min(8, args.num_workers),
                                         pin_memory=True, collate_fn=collate_decoded if args.gpu_preprocess else None,
                                         multiprocessing_context="forkserver" if (args.gpu_preprocess and args.num_workers > 0) else None)
    acc, report, report_dict, cm = calculate_test_accuracy(model, loader, len(test_data), device, _batch_size,
                                                           mode_config_dict['both'], True, image_pipeline=image_pipeline)
    generate_report_and_image(report_dict, acc, cm, "always_both")
    print(test_data.class_to_idx)
    print("Test accuracy random both: {:.2f} %".format(acc))
    print("Test Report:")
    print(report)


if __name__ == '__main__':
    main()
