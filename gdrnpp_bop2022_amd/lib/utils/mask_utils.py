"""COCO run-length strings without pycocotools (lib/utils/mask_utils.py:96-109 `binary_mask_to_rle(compressed=True)`
goes through ``pycocotools.mask.encode``, which is not installed on the target).

``rle_counts_to_string`` / ``rle_string_to_counts`` restate pycocotools' ``rleToString`` / ``rleFrString`` (common/maskApi.c):
every count beyond the third is stored as the difference to the count two places earlier, in 5-bit groups, low group
first, bit 0x20 = "more follows", bit 0x10 of the last group = sign, each character offset by 48."""
from __future__ import annotations

import numpy as np


def rle_counts_to_string(counts) -> str:
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5                       # arithmetic shift: Python ints behave like C's long here
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def rle_string_to_counts(s: str) -> list:
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_from_counts(counts, im_h: int, im_w: int, compressed: bool = True) -> dict:
    """The dict ``binary_mask_to_rle`` returns: {"counts": str | list, "size": [H, W]}."""
    return {"counts": rle_counts_to_string(counts) if compressed else list(counts), "size": [int(im_h), int(im_w)]}


def rle_to_binary_mask(rle: dict) -> np.ndarray:
    counts = rle_string_to_counts(rle["counts"]) if isinstance(rle["counts"], str) else rle["counts"]
    h, w = rle["size"]
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")
