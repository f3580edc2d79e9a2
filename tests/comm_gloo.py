"""Test stand-in for opticommpy_amd.mgpu.RcclComm on boxes without GPUs: the same small interface on top of
torch.distributed's gloo backend.  Test infrastructure only -- the product's communicator is RCCL inside
libssf_hip.so (include/ssf.h: ssf_comm_*)."""
import numpy as np
import torch
import torch.distributed as dist


class GlooComm:
    def __init__(self):
        if not dist.is_initialized():
            dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.calls = []

    @staticmethod
    def _t(a):
        return torch.from_numpy(a.view(np.uint8).reshape(-1))

    def barrier(self):
        dist.barrier()

    def allreduce(self, values, op="sum"):
        t = torch.tensor(np.ascontiguousarray(values, dtype=np.float64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX)
        return t.numpy()

    def bcast(self, arr, root=0):
        self.calls.append(("bcast", arr.nbytes))
        dist.broadcast(self._t(arr), root)
        return arr

    def send(self, arr, peer):
        self.calls.append(("send", peer, arr.nbytes))
        dist.send(self._t(np.ascontiguousarray(arr)), peer)

    def recv(self, arr, peer):
        self.calls.append(("recv", peer, arr.nbytes))
        dist.recv(self._t(arr), peer)
        return arr

    def allgather(self, arr):
        a = np.ascontiguousarray(arr)
        self.calls.append(("allgather", a.nbytes))
        parts = [torch.empty(a.nbytes, dtype=torch.uint8) for _ in range(self.world)]
        dist.all_gather(parts, self._t(a))
        return np.stack([p.numpy().view(a.dtype).reshape(a.shape) for p in parts])

    def close(self):
        dist.barrier()
        dist.destroy_process_group()
