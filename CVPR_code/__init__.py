"""Import-path shim: the reference's callers do ``from CVPR_code.multimodal_model import *`` /
``from CVPR_code.CustomImageTextFolder import *`` (main_both.py:25-27); these modules re-export the HIP-backed
implementations from ``garbage_classification_rca_amd``."""
