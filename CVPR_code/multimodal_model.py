from garbage_classification_rca_amd.multimodal_model import (MM_RCA, EffV2MediumAndDistilbertGated, decision,  # noqa: F401
                                                            HashingTokenizer)
