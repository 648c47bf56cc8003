"""IDRCell100k-format reader (reference: `IDRCell100K`, src/data/custom_datasets.py:153-220).

Format: `<root>/train.csv` (or `test.csv`), one row per image: `image_id, "['rel/path/ch0.tif', 'rel/path/ch1.tif', ...]"`;
channel files live under `<root>/images/`; every file is one single-channel image, a sample has 1-10 of them.
`__getitem__` keeps the reference's contract -- HWC float32 array through `transform(image=...)`, dummy label -1 -- and
`read_planes` hands the raw (C, H, W) planes to the device pipeline (chadavit_amd.data.device_pipeline) instead.
The path list is parsed with `ast.literal_eval` (the reference `eval`s the csv cell)."""
from __future__ import annotations

import ast
import csv
import os
import random
from typing import List, Tuple

import numpy as np


class IDRCell100K:
    def __init__(self, root_dir=None, train=True, transform=None, shuffle=False, sample_ratio=1.0):
        self.root_dir = root_dir
        self.train = train
        self.transform = transform
        self.file_list = self._collect_files()
        self.sample_ratio = sample_ratio
        if shuffle:
            random.shuffle(self.file_list)

    def _collect_files(self) -> List[Tuple[str, List[str]]]:
        self.csv_file = os.path.join(self.root_dir, "train.csv" if self.train else "test.csv")
        out = []
        with open(self.csv_file, "r") as f:
            for row in csv.reader(f):
                image_id, channel_paths = row[0], row[1]
                try:
                    channel_paths = ast.literal_eval(channel_paths)
                except (ValueError, SyntaxError):
                    pass
                if isinstance(channel_paths, str):
                    channel_paths = [channel_paths]
                out.append((image_id, [os.path.join(self.root_dir, "images", p) for p in channel_paths]))
        return out

    def __len__(self):
        return len(self.file_list)

    def num_channels(self) -> List[int]:
        """Channel count per sample (what the token-balanced sampler needs) without opening any image."""
        return [len(paths) for _, paths in self.file_list]

    def read_planes(self, index: int, raw: bool = False) -> np.ndarray:
        """(C, H, W) float32: one plane per channel file, values as stored (custom_datasets.py:181-190).
        raw = True: when every channel file of the sample is 8-bit (or every one 16-bit) unsigned, the planes in THAT type -- the device
        pipeline uploads them as stored and converts on the GPU (same values as the float32 cast, a quarter / half of the bytes)."""
        from PIL import Image
        _, paths = self.file_list[index]
        planes = [np.array(Image.open(p)) for p in paths]
        for p, a in zip(paths, planes):
            if a.ndim != 2:
                raise RuntimeError(f"{p}: expected a single-channel image, got shape {a.shape}")
        if raw and len({a.dtype for a in planes}) == 1 and planes[0].dtype in (np.dtype(np.uint8), np.dtype(np.uint16)):
            return np.stack(planes, 0)
        return np.stack(planes, 0).astype(np.float32)

    def __getitem__(self, index):
        img = np.ascontiguousarray(self.read_planes(index).transpose(1, 2, 0))  # HWC, as the reference builds it
        if self.transform is not None:
            return self.transform(image=img), -1
        return img, -1
