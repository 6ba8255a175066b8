#!/usr/bin/env python3
"""Drop-in `train.py`: every flag of the reference (train.py:27-55) is kept verbatim; the step
loop (reference train.py:94-125) runs on the MI355X TrainEngine -- explicit HIP kernels for
forward/backward, one captured HIP graph per step, fused AdamW, metrics accumulated on the
device -- and shards each global minibatch data-parallel over the GPUs of the node with ONE RCCL
all-reduce of the flat gradient bucket per step.

    python train.py --dataset cifar10 --pos_encoding rope-axial --synthetic
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py \
        --dataset cifar10 --pos_encoding rope-mixed --batch_size 4096 --synthetic

Additions to the reference CLI: --synthetic (CIFAR/MNIST-shaped random batches generated on the
device: torchvision and the dataset downloads are unavailable offline), --steps_per_epoch,
--fp32 (exact-fp32 MFMA instead of the default bf16).  --batch_size is the GLOBAL batch.
"""
import argparse
import csv
import math
import os
import sys
from datetime import datetime

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "vit-rpe-rope_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def get_args(argv=None):
    parser = argparse.ArgumentParser(description='Vision Transformer Training (MI355X)')
    parser.add_argument('--log_dir', type=str, default='logs')
    parser.add_argument('--ckpt_dir', type=str, default='checkpoints')
    parser.add_argument('--dataset', type=str, default='mnist', choices=['mnist', 'cifar10'])
    parser.add_argument('--pos_encoding', type=str, default='absolute',
                        choices=['none', 'absolute', 'relative', 'polynomial', 'rope-axial', 'rope-mixed'])
    parser.add_argument('--rope_theta', type=float, default=100.0)
    parser.add_argument('--poly_degree', type=int, default=3)
    parser.add_argument('--poly_shared_heads', action='store_true', default=True)
    parser.add_argument('--no-poly_shared_heads', action='store_false', dest='poly_shared_heads')
    parser.add_argument('--batch_size', type=int, default=128)
    parser.add_argument('--epochs', type=int, default=25)
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--weight_decay', type=float, default=0.01)
    parser.add_argument('--img_size', type=int, default=32)
    parser.add_argument('--patch_size', type=int, default=4)
    parser.add_argument('--embed_dim', type=int, default=192)
    parser.add_argument('--depth', type=int, default=6)
    parser.add_argument('--num_heads', type=int, default=6)
    # additions
    parser.add_argument('--synthetic', action='store_true', help='dataset-shaped random batches generated on the device')
    parser.add_argument('--steps_per_epoch', type=int, default=0, help='0 = dataset size // batch_size')
    parser.add_argument('--fp32', action='store_true', help='exact-fp32 MFMA instead of bf16')
    parser.add_argument('--data_dir', type=str, default='./data',
                        help='directory holding the dataset in its binary format (CIFAR-10 binary batches / MNIST idx '
                             'files); it is loaded once into HBM as uint8 (no download: there is no network here)')
    parser.add_argument('--seed', type=int, default=0, help='seed of the per-epoch shuffle')
    args = parser.parse_args(argv)
    check_geometry(parser, args)
    return args


def check_geometry(parser, args):
    """The reference accepts any geometry ATen can run; here the attention kernels are compiled for a fixed set, so an
    unsupported combination of flags is refused up front with the list of what works (instead of a launch error)."""
    from vitpe import _lib
    if args.img_size % args.patch_size or args.embed_dim % args.num_heads:
        parser.error(f"--img_size {args.img_size} must be a multiple of --patch_size {args.patch_size} and --embed_dim "
                     f"{args.embed_dim} of --num_heads {args.num_heads}")
    n_tok = (args.img_size // args.patch_size) ** 2 + 1
    hd = args.embed_dim // args.num_heads
    dt = _lib.F32 if args.fp32 else _lib.BF16
    h = _lib.lib()
    if not (h.vitpe_fused_attention_supported(dt, n_tok, args.embed_dim, hd) or h.vitpe_attention_core_supported(dt, n_tok, hd)):
        ok = sorted({n for n in range(2, 300) if h.vitpe_attention_core_supported(dt, n, 32)})
        spans = []
        for n in ok:
            if spans and n == spans[-1][1] + 1:
                spans[-1][1] = n
            else:
                spans.append([n, n])
        parser.error(f"no attention kernel for {n_tok} tokens (img {args.img_size} / patch {args.patch_size}) with head "
                     f"dimension {hd} (--embed_dim {args.embed_dim} / --num_heads {args.num_heads}): supported head "
                     f"dimensions 32 and 64, token counts " + ", ".join(f"{a}-{b}" for a, b in spans) +
                     " (e.g. 32/8 -> 17, 28/4 -> 50, 32/4 -> 65, 224/16 -> 197, 64/4 -> 257)")
    if args.pos_encoding == 'polynomial' and not 0 <= args.poly_degree <= 7:
        parser.error("--poly_degree must be in 0..7 (the attention kernels tabulate the polynomial up to degree 7)")


DATASETS = {'mnist': dict(in_chans=1, num_classes=10, train=60000, test=10000),
            'cifar10': dict(in_chans=3, num_classes=10, train=50000, test=10000)}


class SyntheticBatches:
    """Dataset-shaped batches drawn on the device: images ~ N(0,1) (what Normalize produces,
    reference train.py:72,82), labels ~ U{0..9}; a fixed pool so a model can actually fit it."""

    def __init__(self, n_batches, batch, in_chans, img, num_classes, device, seed):
        g = torch.Generator(device=device).manual_seed(seed)
        pool = max(1, min(n_batches, 8))
        self.images = [torch.randn(batch, in_chans, img, img, generator=g, device=device) for _ in range(pool)]
        self.labels = [torch.randint(0, num_classes, (batch,), generator=g, device=device) for _ in range(pool)]
        self.n = n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.images[i % len(self.images)], self.labels[i % len(self.labels)]


class ResidentBatches:
    """One epoch's index batches over a `vitpe.data.ResidentDataset` (reference DataLoader(batch_size, shuffle=...),
    train.py:89-90, no drop_last): the loop hands the engine sample indices, the pixels never leave HBM.  Yields
    (idx, n_local, n_global) per GLOBAL batch; the ragged last batch is kept, as in the reference."""

    def __init__(self, dataset, global_batch, shuffle, seed, rank, world):
        self.ds, self.global_batch, self.shuffle, self.seed = dataset, global_batch, shuffle, seed
        self.rank, self.world = rank, world
        self.epoch = 0

    def __len__(self):
        return (len(self.ds) + self.global_batch - 1) // self.global_batch

    def __iter__(self):
        from vitpe.data import epoch_global_batches
        it = epoch_global_batches(len(self.ds), self.global_batch, self.epoch, self.seed, self.shuffle, self.rank,
                                  self.world, self.ds.device)
        self.epoch += 1
        return it


def get_dataset(args, info, per_rank_batch, device, rank, world=1):
    if not args.synthetic:
        from vitpe._lib import VitpeError
        from vitpe.data import ResidentDataset
        root = os.path.join(args.data_dir, {'cifar10': 'cifar-10-batches-bin', 'mnist': 'MNIST/raw'}[args.dataset])
        root = root if os.path.isdir(root) else args.data_dir
        try:
            tr = ResidentDataset.from_files(args.dataset, root, True, device, args.img_size)
            te = ResidentDataset.from_files(args.dataset, root, False, device, args.img_size)
        except VitpeError as e:
            raise SystemExit(f"{e}\nno dataset under {args.data_dir} (nothing is downloaded here): "
                             f"place the binary files there or re-run with --synthetic")
        return (ResidentBatches(tr, args.batch_size, True, args.seed, rank, world),
                ResidentBatches(te, args.batch_size, False, args.seed, rank, world))
    n_train = args.steps_per_epoch or info['train'] // args.batch_size
    n_test = max(1, min(n_train // 5, info['test'] // args.batch_size))
    mk = lambda n, seed: SyntheticBatches(n, per_rank_batch, info['in_chans'], args.img_size,  # noqa: E731
                                          info['num_classes'], device, seed)
    return mk(n_train, 1234 + rank), mk(n_test, 4321 + rank)


def train(engine, loader):
    """One epoch (reference train.py:94-125) -> (avg_loss, acc%) over ALL ranks' samples. No per-step host sync."""
    seen = 0
    resident = isinstance(loader, ResidentBatches)
    engine.attach_dataset(loader.ds if resident else None)
    for item in loader:
        if resident:
            idx, n_local, n_global = item
            engine.step_indexed(idx, n_global, n_valid=n_local)
            seen += n_global
        else:
            images, labels = item
            engine.step(images, labels)
            seen += images.shape[0] * engine.world
    loss_sum, correct = engine.read_metrics()      # global-batch means summed over the steps; all ranks' #correct
    return loss_sum / max(len(loader), 1), 100. * correct / max(seen, 1)


def test(engine, loader):
    """Evaluation (reference train.py:127-155): forward only on the same kernels, every sample of the set counted
    (ragged last batch included), totals summed over the ranks."""
    seen = 0
    acc = torch.zeros(2, device=engine.dev)
    resident = isinstance(loader, ResidentBatches)
    if not resident:
        engine.attach_dataset(None)
    for item in loader:
        if resident:
            idx, n_local, n_global = item
            engine.forward_indexed(idx, loader.ds)
            engine.eval_loss(n_local, acc, n_global)
            seen += n_global
        else:
            images, labels = item
            engine.forward_only(images, labels)     # (pads a short batch: images AND labels, like the resident path)
            engine.eval_loss(images.shape[0], acc, images.shape[0] * engine.world)
            seen += images.shape[0] * engine.world
    if engine.world > 1:
        torch.distributed.all_reduce(acc)
    loss_sum, correct = acc.tolist()
    return loss_sum / max(len(loader), 1), 100. * correct / max(seen, 1)


def main(argv=None):
    args = get_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    from vitpe import ddp
    from vitpe.engine import TrainEngine
    from models.vit import VisionTransformer

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    rank, world, _ = ddp.init_from_env(device=device)
    if args.batch_size % world != 0:   # (epoch_global_batches cuts every global batch into `world` equal shares)
        raise SystemExit(f"--batch_size {args.batch_size} is not divisible by the world size {world}")
    lo, hi = ddp.shard_bounds(args.batch_size, rank, world)
    per_rank = hi - lo
    info = DATASETS[args.dataset]

    log_file = None
    if rank == 0:
        os.makedirs(args.log_dir, exist_ok=True)
        os.makedirs(args.ckpt_dir, exist_ok=True)
        timestamp = datetime.now().strftime('%Y%m%d_%H%M%S')
        log_file = f'{args.log_dir}/{args.dataset}_{args.pos_encoding}_{timestamp}.csv'
        with open(log_file, 'w', newline='') as f:
            csv.writer(f).writerow(['epoch', 'train_loss', 'train_acc', 'test_loss', 'test_acc', 'best_acc'])

    train_loader, test_loader = get_dataset(args, info, per_rank, device, rank, world)
    torch.manual_seed(0)
    model = VisionTransformer(img_size=args.img_size, patch_size=args.patch_size, in_chans=info['in_chans'],
                              num_classes=info['num_classes'], embed_dim=args.embed_dim, depth=args.depth,
                              num_heads=args.num_heads, pos_encoding=args.pos_encoding, rope_theta=args.rope_theta,
                              poly_degree=args.poly_degree, poly_shared_heads=args.poly_shared_heads).to(device)
    engine = TrainEngine(model, per_rank, compute_dtype=torch.float32 if args.fp32 else torch.bfloat16,
                         lr=args.lr, weight_decay=args.weight_decay)
    engine.broadcast_parameters(0)

    best_acc = 0
    for epoch in range(args.epochs):
        # CosineAnnealingLR(T_max=epochs), stepped per epoch (reference train.py:196,205)
        engine.set_lr(0.5 * args.lr * (1 + math.cos(math.pi * epoch / args.epochs)))
        train_loss, train_acc = train(engine, train_loader)
        test_loss, test_acc = test(engine, test_loader)
        if rank == 0:
            print(f'\nEpoch: {epoch + 1}/{args.epochs}')
            if test_acc > best_acc:
                best_acc = test_acc
                torch.save(model.state_dict(), f'{args.ckpt_dir}/{args.dataset}_{args.pos_encoding}_best.pth')
            with open(log_file, 'a', newline='') as f:
                csv.writer(f).writerow([epoch + 1, train_loss, train_acc, test_loss, test_acc, best_acc])
            print(f'Train Loss: {train_loss:.4f}, Train Acc: {train_acc:.2f}%')
            print(f'Test Loss: {test_loss:.4f}, Test Acc: {test_acc:.2f}%')
            print(f'Best Test Acc: {best_acc:.2f}%')
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
