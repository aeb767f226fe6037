from garbage_classification_rca_amd.CustomImageTextFolder import *  # noqa: F401,F403
from garbage_classification_rca_amd.CustomImageTextFolder import (CustomImageTextFolder, DatasetFolder,  # noqa: F401
                                                                  custom_make_dataset, find_classes, pre_process_text,
                                                                  IMG_EXTENSIONS, default_loader, pil_loader)
