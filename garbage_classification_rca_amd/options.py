"""Command line of the training / evaluation scripts: every flag and default of the reference's shared parser
(options.py:8-116 -- 33 flags, BooleanOptionalAction pairs included), plus additive flags for this build.
"""
import argparse


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    B = argparse.BooleanOptionalAction
    p.add_argument('--epochs', type=int, default=100, help="number of rounds of training")
    p.add_argument('--dataset_folder_name', type=str, default="", help="dataset folder name in the base location")
    p.add_argument('--dataset_folder_name_val', type=str, default="", help="val dataset folder name in the base location")
    p.add_argument('--lr', type=float, default=0.001, help='learning rate')
    p.add_argument('--image_text_dropout', type=float, default=0.33, help='change of dropping either text or image')
    p.add_argument('--image_prob_dropout', type=float, default=0.7, help='change of dropping image when dropping the modalities')
    p.add_argument('--reg', type=float, default=1e-2, help='regularization rate')
    p.add_argument('--model_dropout', type=float, default=0.6, help='model FC layer dropout')
    p.add_argument('--tl', action=B, default=True, help="Whether to use transfer learning or not")
    p.add_argument('--balance_weights', action=B, default=False, help="Whether to use class balance weights or not")
    p.add_argument('--ft_epochs', type=int, default=15, help='number of fine tuning epochs')
    p.add_argument('--fraction_lr', type=float, default=5, help='value to divide the regular LR for to use in fine tuning')
    p.add_argument('--image_model', type=str, default='b4', help='model name')
    p.add_argument('--text_model', type=str, default='distilbert', help='model name')
    p.add_argument('--model_path', type=str, default="", help='Model file to calculate accuracy against the test set.')
    p.add_argument('--acc_steps', type=int, default=0, help='Gradient accumulation steps')
    p.add_argument('--acc_steps_FT', type=int, default=0, help='Gradient accumulation steps')
    p.add_argument('--num_neurons_FC', type=int, default=256, help='Num neurons in FC layers')
    p.add_argument('--batch_size', type=int, default=16, help='Batch size')
    p.add_argument('--batch_size_FT', type=int, default=16, help='Batch size for fine tuning')
    p.add_argument('--opt', type=str, default="sgd", help='Optimizer to use')
    p.add_argument('--base_path', type=str, default=r"D:\Mestrado\ENSF_619_02_Final_project_jose_cazarin\BEST_MODELS_CVPR_2025", help='base_path')
    p.add_argument('--calculate_dataset_stats', action=B, default=False, help="Calculate the development set stats used for normalization")
    p.add_argument('--prob_aug', type=float, default=0.6, help='Probability of applying augmentations')
    p.add_argument('--late_fusion', type=str, default="gated", help='Which late fusion strategy to use')
    p.add_argument('--label_smoothing', type=float, default=0.0, help='Fraction to use Label Smoothing')
    p.add_argument('--name', type=str, help='Run description')
    p.add_argument('--reverse', action=B, default=False, help="Use RCA or not")
    p.add_argument('--features_only', action=B, default=False, help="Use only the extracted features or not")
    p.add_argument('--cross_attention_only', action=B, default=False, help="Use only the cross attention features or not")
    p.add_argument('--extended_desc_train', type=str, help='Path to extended description train CSV file')
    p.add_argument('--extended_desc_val', type=str, help='Path to extended description val CSV file')
    p.add_argument('--balanced_sampler', action=B, default=False, help="Use balanced sampler or not")
    p.add_argument('--use_synonyms', action=B, default=False, help="Use synonymizer augmentation for text")
    p.add_argument('--prob_aug_text', type=float, default=0.6, help='Prob of applying text synonymization augmentations')
    p.add_argument('--classifier_weights', type=str, help='Path to weights file of the classifier head in the Q-Former model')
    # ---- additive flags of this build (none of the above changed meaning) ----
    p.add_argument('--tokens_max_len', type=int, default=None, help='caption length (default: the text model maximum, as the reference)')
    p.add_argument('--dtype', type=str, default="bf16", choices=["bf16", "fp32"], help='compute dtype of the HIP path')
    p.add_argument('--synthetic', type=int, default=0, help='>0: train on this many synthetic (image, caption) pairs instead of a folder')
    p.add_argument('--num_workers', type=int, default=16, help='DataLoader workers (reference: 16)')
    p.add_argument('--seed', type=int, default=None, help='seed torch/numpy (reference leaves this commented out)')
    return p


def args_parser(argv=None):
    return build_parser().parse_args(argv)
