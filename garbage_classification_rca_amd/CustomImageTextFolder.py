"""Image + caption folder dataset with the API of the reference's CVPR_code/CustomImageTextFolder.py:379-467
(``DatasetFolder`` :145-346, ``custom_make_dataset`` :45-126, ``find_classes`` :130-142, ``pre_process_text`` :29-42),
without the torchvision dependency.

Layout: ``root/<class>/**/<caption words>.<image ext>``; the caption of a sample is its cleaned file stem.  Items are
``({'image': {'raw_image', 'image_path'}, 'text': {'original_text', 'tokens', 'attention_mask'}}, target)``.
"""
from __future__ import annotations

import os
import re
import sys
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
from PIL import Image

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")


def has_file_allowed_extension(filename: str, extensions: Union[str, Tuple[str, ...]]) -> bool:
    return filename.lower().endswith(extensions if isinstance(extensions, str) else tuple(extensions))


def pre_process_text(text: str) -> str:
    """lower -> '_' to space -> drop digits -> drop everything but [a-zA-Z ] -> strip  (reference :29-42)."""
    text = text.lower().replace("_", " ")
    text = re.sub(r"[0-9]", "", text)
    text = re.sub(r"[^a-zA-Z ]+", "", text)
    return text.strip()


def find_classes(directory: str) -> Tuple[List[str], Dict[str, int]]:
    classes = sorted(e.name for e in os.scandir(directory) if e.is_dir())
    if not classes:
        raise FileNotFoundError(f"Couldn't find any class folder in {directory}.")
    return classes, {c: i for i, c in enumerate(classes)}


def custom_make_dataset(directory: str, extended_desc, class_to_idx: Optional[Dict[str, int]] = None,
                        extensions: Optional[Union[str, Tuple[str, ...]]] = None,
                        is_valid_file: Optional[Callable[[str], bool]] = None):
    """Returns (per_class_lists, instances).  ``per_class_lists`` has exactly four lists, as the reference (:94)."""
    directory = os.path.expanduser(directory)
    if class_to_idx is None:
        _, class_to_idx = find_classes(directory)
    elif not class_to_idx:
        raise ValueError("'class_to_index' must have at least one entry to collect any samples.")
    if (extensions is None) == (is_valid_file is None):
        raise ValueError("Both extensions and is_valid_file cannot be None or not None at the same time")
    if extensions is not None:
        def is_valid_file(x: str) -> bool:  # noqa: F811
            return has_file_allowed_extension(x, extensions)
    lookup = None
    if extended_desc is not None:
        try:
            import pandas as pd
            df = pd.read_csv(extended_desc, dtype=str)
            lookup = df.set_index("filename")["description"]
        except Exception as e:
            print(f"Error reading {extended_desc}: {e}", file=sys.stderr)
            sys.exit(1)
    available = set()
    per_class_lists: List[list] = [[], [], [], []]
    instances = []
    for target_class in sorted(class_to_idx.keys()):
        class_index = class_to_idx[target_class]
        target_dir = os.path.join(directory, target_class)
        if not os.path.isdir(target_dir):
            continue
        for root, _, fnames in sorted(os.walk(target_dir, followlinks=True)):
            for fname in sorted(fnames):
                path = os.path.join(root, fname)
                if not is_valid_file(path):
                    continue
                p = Path(path)
                long_desc = ""
                if lookup is not None:
                    long_desc = lookup.get(os.path.join(p.parent.name, p.name))
                item = ({"text": pre_process_text(p.stem), "image": path, "long_text": long_desc}, class_index)
                per_class_lists[class_index].append(item)
                instances.append(item)
                available.add(target_class)
    empty = set(class_to_idx.keys()) - available
    if empty:
        msg = f"Found no valid file for the classes {', '.join(sorted(empty))}. "
        if extensions is not None:
            msg += f"Supported extensions are: {extensions if isinstance(extensions, str) else ', '.join(extensions)}"
        raise FileNotFoundError(msg)
    return per_class_lists, instances


def pil_loader(path: str) -> Image.Image:
    with open(path, "rb") as f:
        return Image.open(f).convert("RGB")


default_loader = pil_loader


def _tokenize(tokenizer, text, max_len):
    """The reference calls ``tokenizer.encode_plus`` (:305-333); transformers >= 5 removed it, the plain call takes the
    same arguments."""
    fn = getattr(tokenizer, "encode_plus", None) or tokenizer
    return fn(text, max_length=max_len, truncation=True, return_attention_mask=True, return_token_type_ids=False,
              padding="max_length", return_tensors="pt")


class DatasetFolder(torch.utils.data.Dataset):
    def __init__(self, root: str, tokens_max_len: int, tokenizer_text, loader: Callable[[str], Any], extended_desc: str,
                 extensions: Optional[Tuple[str, ...]] = None, transform: Optional[Callable] = None,
                 target_transform: Optional[Callable] = None, is_valid_file: Optional[Callable[[str], bool]] = None,
                 list_custom_samples=None) -> None:
        self.root, self.transform, self.target_transform = root, transform, target_transform
        classes, class_to_idx = self.find_classes(self.root)
        custom_samples = ()
        if self.root is not None:
            custom_samples = custom_make_dataset(self.root, extended_desc, class_to_idx, extensions, is_valid_file)
        self.per_class = None if root is None else custom_samples[0]
        self.loader, self.extensions = loader, extensions
        self.classes, self.class_to_idx = classes, class_to_idx
        self.tokens_max_len, self.tokenizer, self.extended_desc = tokens_max_len, tokenizer_text, extended_desc
        if list_custom_samples is not None:
            self.samples = list_custom_samples
        else:
            self.samples = custom_samples[1]
        self.targets = [s[1] for s in self.samples]

    def find_classes(self, directory: str):
        return find_classes(directory)

    def __getitem__(self, index: int):
        path, target = self.samples[index]
        sample_image = self.loader(path["image"])
        if self.transform is not None:
            sample_image = self.transform(sample_image)
        if self.target_transform is not None:
            target = self.target_transform(target)
        text_key = "long_text" if self.extended_desc is not None else "text"
        tokens_dict = {"original_text": path[text_key]}
        if self.tokenizer is not None:
            # pre-tokenised caption cache: a caption is tokenised once per worker, not once per
            # epoch -- the result depends only on (caption, max_len), the reference re-runs encode_plus every time (:305-336)
            cache = self.__dict__.setdefault("_token_cache", {})
            hit = cache.get(path[text_key])
            if hit is None:
                enc = _tokenize(self.tokenizer, path[text_key], self.tokens_max_len)
                hit = (enc["input_ids"].flatten(), enc["attention_mask"].flatten())
                cache[path[text_key]] = hit
            tokens_dict["tokens"], tokens_dict["attention_mask"] = hit[0].clone(), hit[1].clone()
        return {"image": {"raw_image": sample_image, "image_path": path["image"]}, "text": tokens_dict}, target

    def __len__(self) -> int:
        return len(self.samples)


class CustomImageTextFolder(DatasetFolder):
    def __init__(self, root: str, tokens_max_len=None, tokenizer_text: Optional[Callable] = None, custom_samples=None,
                 transform: Optional[Callable] = None, target_transform: Optional[Callable] = None,
                 loader: Callable[[str], Any] = default_loader, is_valid_file: Optional[Callable[[str], bool]] = None,
                 extended_desc=None):
        super().__init__(root, tokens_max_len, tokenizer_text, loader, extended_desc,
                         IMG_EXTENSIONS if is_valid_file is None else None, transform=transform,
                         target_transform=target_transform, is_valid_file=is_valid_file, list_custom_samples=custom_samples)
        self.imgs = self.samples

    def get_tokens(self, text_data):
        enc = _tokenize(self.tokenizer, text_data, self.tokens_max_len)
        return enc["input_ids"].flatten(), enc["attention_mask"].flatten()


class SyntheticImageTextDataset(torch.utils.data.Dataset):
    """Synthetic pairs of SURVEY.md section 8(d) with the item structure of CustomImageTextFolder: N(0,1) images
    (stand for ImageNet-normalised pixels), captions [CLS] ids [SEP] pad with lengths U[8, S], labels i mod 4."""

    def __init__(self, n, image_size, tokens_max_len, seed_images=1234, seed_text=4321, vocab=30522, n_classes=4):
        from .procedural import synth_captions
        self.n, self.image_size, self.seed_images = n, image_size, seed_images
        self.ids, self.mask = (torch.from_numpy(a) for a in synth_captions(n, tokens_max_len, seed_text, vocab_hi=vocab))
        self.targets = [i % n_classes for i in range(n)]
        self.per_class = [[i for i in range(n) if i % n_classes == c] for c in range(n_classes)]
        self.classes = ["Black", "Blue", "Green", "TTR"][:n_classes]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed_images * 1000003 + i)
        img = torch.randn(3, self.image_size, self.image_size, generator=g)
        return ({"image": {"raw_image": img, "image_path": f"synthetic://{i}"},
                 "text": {"original_text": "", "tokens": self.ids[i], "attention_mask": self.mask[i]}}, self.targets[i])
