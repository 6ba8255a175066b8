"""GPU-resident input pipeline: the uint8 dataset lives in HBM, a step's samples are gathered,
converted and normalised inside the patch-embed unfold kernel (vitpe_unfold_u8).  Replaces the reference's
DataLoader + transforms.{Resize, ToTensor, Normalize} (train.py:58-92); shuffling follows DataLoader(shuffle=True):
a fresh random permutation per epoch, here drawn on the device from a seeded generator and sharded over ranks
in equal contiguous slices of the permutation.

Readers for the datasets' own binary formats (no pickle): CIFAR-10 "binary version" (data_batch_{1..5}.bin,
test_batch.bin: 1 label byte + 3072 pixel bytes, channel-planar) and MNIST idx files
({train,t10k}-images-idx3-ubyte, -labels-idx1-ubyte; 28x28 is resized to 32x32 once at load time).
"""
from __future__ import annotations

import os
import struct
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from ._lib import VitpeError

DATASET_STATS = {   # transforms.Normalize arguments of reference train.py:72,81
    "mnist": ((0.1307,), (0.3081,)),
    "cifar10": ((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010)),
}


def read_cifar10_bin(root: str, train: bool) -> Tuple[np.ndarray, np.ndarray]:
    """-> (images uint8 [N,3,32,32], labels int64 [N]) from the CIFAR-10 binary batches under `root`."""
    names = [f"data_batch_{i}.bin" for i in range(1, 6)] if train else ["test_batch.bin"]
    imgs, labs = [], []
    for n in names:
        path = os.path.join(root, n)
        if not os.path.exists(path):
            raise VitpeError(f"{path} not found (expected the CIFAR-10 binary version)")
        raw = np.fromfile(path, dtype=np.uint8)
        if raw.size % 3073 != 0:
            raise VitpeError(f"{path}: size {raw.size} is not a multiple of 3073-byte records")
        rec = raw.reshape(-1, 3073)
        labs.append(rec[:, 0].astype(np.int64))
        imgs.append(rec[:, 1:].reshape(-1, 3, 32, 32))
    return np.concatenate(imgs), np.concatenate(labs)


def _read_idx(path: str) -> np.ndarray:
    if not os.path.exists(path):
        raise VitpeError(f"{path} not found (expected the MNIST idx files)")
    with open(path, "rb") as f:
        magic = struct.unpack(">I", f.read(4))[0]
        if magic >> 8 != 0x08:   # 0x0000 08 <ndim>: unsigned byte data
            raise VitpeError(f"{path}: not an idx ubyte file (magic {magic:#x})")
        dims = struct.unpack(">" + "I" * (magic & 0xFF), f.read(4 * (magic & 0xFF)))
        data = np.frombuffer(f.read(), dtype=np.uint8)
    if data.size != int(np.prod(dims)):
        raise VitpeError(f"{path}: {data.size} bytes for dims {dims}")
    return data.reshape(dims)


def read_mnist_idx(root: str, train: bool, img_size: int = 32) -> Tuple[np.ndarray, np.ndarray]:
    """-> (images uint8 [N,1,S,S], labels int64 [N]).  transforms.Resize(img_size) (bilinear, on the uint8 image,
    train.py:70) is applied once here; torchvision / PIL are absent from the build image, so the resize is torch's
    bilinear (half-pixel centres) rounded to uint8 -- parity with PIL's fixed-point resize is unpinned (+-1 LSB)."""
    pre = "train" if train else "t10k"
    x = _read_idx(os.path.join(root, f"{pre}-images-idx3-ubyte"))
    y = _read_idx(os.path.join(root, f"{pre}-labels-idx1-ubyte")).astype(np.int64)
    x = torch.from_numpy(x.copy()).unsqueeze(1).float()
    if x.shape[-1] != img_size:
        x = torch.nn.functional.interpolate(x, size=(img_size, img_size), mode="bilinear", align_corners=False)
    return x.round().clamp(0, 255).to(torch.uint8).numpy(), y


class ResidentDataset:
    """uint8 images [N,C,S,S] + int64 labels [N] held on one device, with the normalisation constants."""

    def __init__(self, images_u8, labels, mean, std, device="cuda"):
        images_u8 = torch.as_tensor(images_u8)
        labels = torch.as_tensor(labels)
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[2] != images_u8.shape[3]:
            raise VitpeError("ResidentDataset: images must be uint8 [N,C,S,S]")
        if labels.shape != (images_u8.shape[0],):
            raise VitpeError("ResidentDataset: one label per image")
        if len(mean) != images_u8.shape[1] or len(std) != images_u8.shape[1]:
            raise VitpeError("ResidentDataset: one mean/std per channel")
        self.device = torch.device(device)
        self.images = images_u8.contiguous().to(self.device)
        self.labels = labels.to(torch.int64).to(self.device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)

    def __len__(self):
        return self.images.shape[0]

    @classmethod
    def from_files(cls, dataset: str, root: str, train: bool, device="cuda", img_size: int = 32):
        if dataset == "cifar10":
            x, y = read_cifar10_bin(root, train)
            if img_size != 32:
                raise VitpeError("cifar10 binary records are 32x32; other --img_size values need a resize pass")
        elif dataset == "mnist":
            x, y = read_mnist_idx(root, train, img_size)
        else:
            raise VitpeError(f"unknown dataset {dataset}")
        mean, std = DATASET_STATS[dataset]
        return cls(x, y, mean, std, device)


def shard_slice(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Equal contiguous shard [lo, hi) of a (permuted) index list; the tail n_items % world is dropped so that all
    ranks run the same number of steps (the gradient all-reduce needs every rank in every step)."""
    per = n_items // world
    return rank * per, (rank + 1) * per


def epoch_batches(n_items: int, batch: int, epoch: int, seed: int = 0, shuffle: bool = True, rank: int = 0,
                  world: int = 1, device="cuda", drop_last: bool = True) -> Iterator[torch.Tensor]:
    """Index batches (int64 on `device`) of one epoch for this rank.  All ranks draw the same permutation
    (generator seeded with seed + epoch) and take their own slice."""
    dev = torch.device(device)
    if shuffle:
        g = torch.Generator(device=dev)
        g.manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g, device=dev)
    else:
        order = torch.arange(n_items, device=dev)
    lo, hi = shard_slice(n_items, rank, world)
    mine = order[lo:hi]
    nfull = mine.numel() // batch
    for i in range(nfull):
        yield mine[i * batch:(i + 1) * batch]
    if not drop_last and mine.numel() % batch:
        yield mine[nfull * batch:]


def epoch_global_batches(n_items: int, global_batch: int, epoch: int, seed: int = 0, shuffle: bool = True, rank: int = 0,
                         world: int = 1, device="cuda") -> Iterator[Tuple[torch.Tensor, int, int]]:
    """The reference's DataLoader(batch_size=global_batch, shuffle=..., drop_last=False) (train.py:89-90) seen from
    one rank: every GLOBAL batch k = order[k*G : (k+1)*G] (the last one ragged: 50000 % 128 = 80, 10000 % 128 = 16) is
    cut into `world` contiguous shares of G / world slots; this rank gets the indices that fall into its share.
    Yields (idx, n_local, n_global): idx int64 [max(n_local, 1)] on `device` (a rank whose share of a ragged batch is
    empty still gets one index so that it runs the step -- the all-reduce needs every rank -- with n_local = 0),
    n_global = size of the global batch.  Every sample of the dataset is visited exactly once per epoch and all ranks
    make ceil(n_items / G) steps."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    dev = torch.device(device)
    if shuffle:
        g = torch.Generator(device=dev)
        g.manual_seed(seed + epoch)
        order = torch.randperm(n_items, generator=g, device=dev)
    else:
        order = torch.arange(n_items, device=dev)
    per = global_batch // world
    for k0 in range(0, n_items, global_batch):
        n_global = min(global_batch, n_items - k0)
        lo = min(rank * per, n_global)
        hi = min((rank + 1) * per, n_global)
        if hi > lo:
            yield order[k0 + lo:k0 + hi], hi - lo, n_global
        else:
            yield order[k0:k0 + 1], 0, n_global
