"""child of tests/test_00_bench_child.py::test_ipc_done_mask_exchange_between_two_processes_on_one_gpu: run under
torch.distributed.run with 2+ ranks that all use cuda:0 (gloo for the rendezvous; RCCL refuses duplicate devices, IPC
does not).  Every rank publishes blocks whose content names (rank, block, row, word); after each fence every rank
checks every other rank's blocks -- with NO barrier behind the check: the blocks of a window stay readable until the
fence after next (the receive buffers hold two windows), so a slow reader and a fast publisher do not meet."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from aquaticgymenv_amd.sharded import DoneMaskExchange


def block(rank, k, steps, words, dev):
    base = torch.arange(steps * words, dtype=torch.int64, device=dev).reshape(steps, words)
    return base + (rank + 1) * 1000003 + k * 7919


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    steps, words, slots = 20, 512, 3
    ex = DoneMaskExchange(steps, words, dev, kind="ipc", slots=slots)
    work = torch.cuda.Stream(device=dev)
    k = 0
    for rnd in range(6):
        used = []
        with torch.cuda.stream(work):
            count = slots if rnd % 2 == 0 else 1
            for j in range(count):
                src = block(rank, k, steps, words, dev)
                # the round's last block goes out as ONE fan-out launch on the producing stream (final=True), the
                # others through the pump thread and the per-peer side streams
                used.append((ex.gather_async(src, source_id=j, final=(j == count - 1 and rnd >= 1)), k))
                k += 1
        work.synchronize()                  # final=True leaves the producing stream to the caller: drained before the fence
        ex.fence()
        if rank == 0 and rnd % 2 == 1:
            import time
            time.sleep(0.3)                 # a slow reader: the others are already publishing the NEXT window meanwhile --
        for slot, kk in used:               # into the other half of the receive buffers, so no barrier is needed here
            for r in range(world):
                got = ex.gathered[slot][r]
                assert torch.equal(got, block(r, kk, steps, words, dev)), "rank %d: block %d of rank %d is wrong" % (rank, kk, r)
    # more blocks than a window holds between two fences must be refused, not silently overwrite a peer's unread block
    try:
        for j in range(slots + 1):
            ex.gather_async(block(rank, 0, steps, words, dev))
        raise SystemExit("rank %d: the %d-th block since the fence was accepted" % (rank, slots + 1))
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    ex.close()
    dist.barrier()
    if rank == 0:
        print("ipc exchange ok: %d ranks on one GPU, %d blocks each" % (world, k))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
