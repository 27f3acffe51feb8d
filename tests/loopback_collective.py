"""TEST INFRASTRUCTURE: P logical ranks as P threads on ONE GPU ("loopback shards").  Each rank has its
own libkrylov_hip context / HIP stream / row shard; the collective below implements all-reduce and
the ghost exchange through shared device tensors and thread barriers.  This exercises the hook-based
row-sharded path (every all-reduce site, the halo exchange, identical host control flow on all ranks)
on the single-GPU box, where RCCL cannot place two ranks on one device."""
import threading

import torch


class LoopbackWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.objs = [None] * world
        self.total = None


class LoopbackCollective:
    def __init__(self, shared: LoopbackWorld, rank: int):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _sync(self):
        torch.cuda.current_stream().synchronize()

    def all_reduce(self, t):
        sh = self.sh
        self._sync()
        sh.slots[self.rank] = t
        sh.barrier.wait()
        if self.rank == 0:
            tot = sh.slots[0].clone()
            for q in range(1, self.world):   # fixed order -> identical bits on every rank
                tot += sh.slots[q]
            sh.total = tot
            self._sync()
        sh.barrier.wait()
        t.copy_(sh.total)
        self._sync()
        sh.barrier.wait()

    def all_gather_object(self, obj):
        sh = self.sh
        sh.objs[self.rank] = obj
        sh.barrier.wait()
        out = list(sh.objs)
        sh.barrier.wait()
        return out

    def exchange(self, sendbuf, send_counts, recvbuf, recv_counts):
        sh = self.sh
        self._sync()
        sh.slots[self.rank] = (sendbuf, list(send_counts))
        sh.barrier.wait()
        ro = 0
        for q in range(self.world):
            n = recv_counts[q]
            if n:
                sb, sc = sh.slots[q]
                so = sum(sc[:self.rank])          # q's send segment destined for me
                assert sc[self.rank] == n
                recvbuf[ro:ro + n].copy_(sb[so:so + n])
                ro += n
        self._sync()
        sh.barrier.wait()
