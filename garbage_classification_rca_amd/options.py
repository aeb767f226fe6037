"""Command line of the training / evaluation scripts.

The flag NAMES, TYPES and DEFAULTS are the reference's shared parser (options.py:8-116: 33 flags, the boolean ones as
``--flag / --no-flag`` pairs) -- that is the interface a reference launch line relies on and what
tests/golden/options_goldens.json pins.  The table form and the descriptions are this build's own.
"""
import argparse

_INT, _FLOAT, _STR, _BOOL = "int", "float", "str", "bool"

# (flag, kind, default, description)
REFERENCE_FLAGS = [
    ("epochs", _INT, 100, "epochs of the frozen-backbone phase"),
    ("dataset_folder_name", _STR, "", "training split: folder under the working directory (the reference joins it to its script directory, main_both.py:39,458; --base_path is unused there too)"),
    ("dataset_folder_name_val", _STR, "", "validation split: folder under the working directory"),
    ("lr", _FLOAT, 0.001, "learning rate of the first phase"),
    ("image_text_dropout", _FLOAT, 0.33, "probability that a training step zeroes one modality"),
    ("image_prob_dropout", _FLOAT, 0.7, "given that a modality is zeroed: probability that it is the image"),
    ("reg", _FLOAT, 1e-2, "weight decay"),
    ("model_dropout", _FLOAT, 0.6, "dropout in front of the classifier"),
    ("tl", _BOOL, True, "start with frozen backbones (transfer learning)"),
    ("balance_weights", _BOOL, False, "class-balanced loss weights"),
    ("ft_epochs", _INT, 15, "epochs of the fine-tuning phase"),
    ("fraction_lr", _FLOAT, 5, "fine-tuning learning rate = lr / fraction_lr"),
    ("image_model", _STR, "b4", "image backbone"),
    ("text_model", _STR, "distilbert", "text backbone"),
    ("model_path", _STR, "", "checkpoint evaluated by the test-split script"),
    ("acc_steps", _INT, 0, "micro-batches per optimizer step, first phase (0: every batch)"),
    ("acc_steps_FT", _INT, 0, "micro-batches per optimizer step, fine-tuning phase"),
    ("num_neurons_FC", _INT, 256, "width of the (unused by MM_RCA) fully connected layers"),
    ("batch_size", _INT, 16, "batch size, first phase"),
    ("batch_size_FT", _INT, 16, "batch size, fine-tuning phase"),
    ("opt", _STR, "sgd", "sgd | adamw"),
    ("base_path", _STR, r"D:\Mestrado\ENSF_619_02_Final_project_jose_cazarin\BEST_MODELS_CVPR_2025", "accepted for CLI compatibility; unused here as in the reference (paths are relative to the working directory)"),
    ("calculate_dataset_stats", _BOOL, False, "recompute the normalisation statistics"),
    ("prob_aug", _FLOAT, 0.6, "probability of each image augmentation"),
    ("late_fusion", _STR, "gated", "fusion head (MM_RCA is the one built here)"),
    ("label_smoothing", _FLOAT, 0.0, "label smoothing of the loss"),
    ("name", _STR, None, "free-text run description"),
    ("reverse", _BOOL, False, "reverse cross-attention, (1 - A) / (n - 1)"),
    ("features_only", _BOOL, False, "classifier sees the backbone features only"),
    ("cross_attention_only", _BOOL, False, "classifier sees the cross-attention outputs only"),
    ("extended_desc_train", _STR, None, "CSV of long captions, training split"),
    ("extended_desc_val", _STR, None, "CSV of long captions, validation split"),
    ("balanced_sampler", _BOOL, False, "class-balanced batch sampling"),
    ("use_synonyms", _BOOL, False, "synonym augmentation of captions"),
    ("prob_aug_text", _FLOAT, 0.6, "probability of the caption augmentation"),
    ("classifier_weights", _STR, None, "classifier-head weights of the Q-Former model"),
]

# additive flags of this build (none of the above changed meaning)
BUILD_FLAGS = [
    ("tokens_max_len", _INT, None, "caption length (default: the text model's maximum, as in the reference)"),
    ("synthetic", _INT, 0, ">0: train on this many synthetic pairs instead of a folder"),
    ("num_workers", _INT, 16, "DataLoader workers"),
    ("seed", _INT, None, "seed torch / numpy"),
    ("image_size", _INT, None, "input side of the conv image backbones (default: 480 for EfficientNetV2-M as in the reference, else 224)"),
    ("blip2_checkpoint", _STR, None, "q_former_training.py: state_dict file (.pth / .safetensors) of Blip2ForConditionalGeneration (no network here: the reference downloads Salesforce/blip2-opt-2.7b); unset = random weights, with a warning"),
    ("hip_graph", _STR, "auto", "auto | on | off: replay each train step's forward/backward launches from a HIP graph captured once per batch shape (training.GraphedTrainStep); auto = one GPU, batch size <= 16 and images up to 224 x 224, where the step is launch-bound"),
    ("gpu_preprocess", _BOOL, True, "image pipeline (pad / resize / augment / normalise) as one GPU launch per batch instead of per-sample CPU transforms"),
]


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    cast = {_INT: int, _FLOAT: float, _STR: str}
    for flag, kind, default, text in REFERENCE_FLAGS + BUILD_FLAGS:
        if kind == _BOOL:
            p.add_argument("--" + flag, action=argparse.BooleanOptionalAction, default=default, help=text)
        else:
            p.add_argument("--" + flag, type=cast[kind], default=default, help=text)
    p.add_argument("--dtype", type=str, default="bf16x3f", choices=["bf16", "fp32", "bf16x3", "bf16x3f"],
                   help="compute mode of the HIP path: bf16x3f (default: forward as bf16x3 -- logits within 1e-3 of the reference's fp32 arithmetic -- "
                        "with the bf16 mode's backward: the fastest mode that meets that bound) | bf16 (fastest; logits ~1e-2 from fp32: opt-in) | bf16x3 (the reference's fp32 arithmetic to "
                        "~1e-5 on logits AND gradients: fp32 storage, every nn.Linear as a three-pass split-bf16 product) | fp32 (every GEMM on "
                        "the fp32 matrix cores)")
    return p


def args_parser(argv=None):
    return build_parser().parse_args(argv)
