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


def _long_caption_table(csv_path):
    """``filename -> description`` of the extended-description CSV (reference :60-68); a broken file ends the run."""
    if csv_path is None:
        return None
    try:
        import pandas as pd
        return pd.read_csv(csv_path, dtype=str).set_index("filename")["description"]
    except Exception as exc:                                     # the reference exits here as well
        print(f"Error reading {csv_path}: {exc}", file=sys.stderr)
        sys.exit(1)


def _files_below(folder: str):
    """every file under ``folder`` (links followed), directories and names in sorted order -- the reference's walk order"""
    for here, _dirs, names in sorted(os.walk(folder, followlinks=True)):
        for name in sorted(names):
            yield os.path.join(here, name)


def custom_make_dataset(directory: str, extended_desc, class_to_idx: Optional[Dict[str, int]] = None,
                        extensions: Optional[Union[str, Tuple[str, ...]]] = None,
                        is_valid_file: Optional[Callable[[str], bool]] = None):
    """(per_class_lists, instances): one ``({'text', 'image', 'long_text'}, class_index)`` per accepted file, classes in
    name order, files in walk order.  ``per_class_lists`` has exactly FOUR lists, as the reference (:94) -- the path
    assumes the four garbage classes."""
    directory = os.path.expanduser(directory)
    if class_to_idx is None:
        class_to_idx = find_classes(directory)[1]
    elif len(class_to_idx) == 0:
        raise ValueError("'class_to_index' must have at least one entry to collect any samples.")
    if (extensions is None) == (is_valid_file is None):
        raise ValueError("Both extensions and is_valid_file cannot be None or not None at the same time")
    accept = is_valid_file if extensions is None else (lambda f: has_file_allowed_extension(f, extensions))
    captions = _long_caption_table(extended_desc)
    per_class_lists: List[list] = [[] for _ in range(4)]
    instances, seen = [], set()
    for cls_name in sorted(class_to_idx):
        folder = os.path.join(directory, cls_name)
        if not os.path.isdir(folder):
            continue
        for file_path in _files_below(folder):
            if not accept(file_path):
                continue
            fp = Path(file_path)
            long_text = "" if captions is None else captions.get(os.path.join(fp.parent.name, fp.name))
            entry = ({"text": pre_process_text(fp.stem), "image": file_path, "long_text": long_text}, class_to_idx[cls_name])
            per_class_lists[class_to_idx[cls_name]].append(entry)
            instances.append(entry)
            seen.add(cls_name)
    missing = sorted(set(class_to_idx) - seen)
    if missing:
        hint = "" if extensions is None else (" Supported extensions are: " + (extensions if isinstance(extensions, str) else ", ".join(extensions)))
        raise FileNotFoundError(f"Found no valid file for the classes {', '.join(missing)}." + hint)
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
        self.loader, self.extensions = loader, extensions
        self.tokens_max_len, self.tokenizer, self.extended_desc = tokens_max_len, tokenizer_text, extended_desc
        self.classes, self.class_to_idx = self.find_classes(root)
        listed = custom_make_dataset(root, extended_desc, self.class_to_idx, extensions, is_valid_file) if root is not None else None
        self.per_class = None if listed is None else listed[0]
        # a caller may hand in its own (sub)list of samples, e.g. a balanced resampling of ``per_class`` (reference :196-201)
        self.samples = list_custom_samples if list_custom_samples is not None else listed[1]
        self.targets = [cls for _, cls in self.samples]

    def find_classes(self, directory: str):
        return find_classes(directory)

    def __getitem__(self, index: int):
        record, target = self.samples[index]
        image = self.loader(record["image"])
        if self.transform is not None:
            image = self.transform(image)
        if self.target_transform is not None:
            target = self.target_transform(target)
        caption = record["long_text" if self.extended_desc is not None else "text"]
        text = {"original_text": caption}
        if self.tokenizer is not None:
            # pre-tokenised caption cache: a caption is tokenised once per worker, not once per
            # epoch -- the result depends only on (caption, max_len), the reference re-runs encode_plus every time (:305-336)
            cache = self.__dict__.setdefault("_token_cache", {})
            hit = cache.get(caption)
            if hit is None:
                enc = _tokenize(self.tokenizer, caption, self.tokens_max_len)
                hit = cache[caption] = (enc["input_ids"].flatten(), enc["attention_mask"].flatten())
            text["tokens"], text["attention_mask"] = hit[0].clone(), hit[1].clone()
        return {"image": {"raw_image": image, "image_path": record["image"]}, "text": text}, target

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
