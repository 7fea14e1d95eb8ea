"""Feasibility probe: stream wait-value / write-value on hipIpc-shared device memory between two processes
(one GPU).  Prints per-call status and a ping-pong round-trip time."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def hiplib():
    import torch

    return C.CDLL(str(Path(torch.__file__).resolve().parent / "lib" / "libamdhip64.so"), mode=C.RTLD_GLOBAL)


def chk(hip, rc, what):
    if rc != 0:
        hip.hipGetErrorString.restype = C.c_char_p
        print(f"[{os.getpid()}] {what}: rc={rc} {hip.hipGetErrorString(rc).decode()}", flush=True)
    return rc


class Handle(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]


def worker(rank, q_in, q_out, rounds):
    import torch

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    hip = hiplib()
    hip.hipStreamWaitValue64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint64]
    hip.hipStreamWriteValue64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipIpcGetMemHandle.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    mine = C.c_void_p()
    chk(hip, hip.hipMalloc(C.byref(mine), 1 << 20), "hipMalloc")
    chk(hip, hip.hipMemset(mine, 0, 1 << 20), "memset")
    h = Handle()
    chk(hip, hip.hipIpcGetMemHandle(C.byref(h), mine), "IpcGetMemHandle")
    q_out.put(bytes(h))
    other = q_in.get()
    ho = Handle.from_buffer_copy(other)
    peer = C.c_void_p()
    chk(hip, hip.hipIpcOpenMemHandle(C.byref(peer), ho, 1), "IpcOpenMemHandle")
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    can = C.c_int()
    hip.hipDeviceGetAttribute(C.byref(can), 10016, 0)  # may be the wrong enum; informational only
    # --- one-shot: rank 1 fills rank 0's data then raises rank 0's flag; rank 0 waits on the flag in-stream
    flag_mine, data_mine = mine.value, mine.value + 4096
    flag_peer, data_peer = peer.value, peer.value + 4096
    if rank == 0:
        rc = chk(hip, hip.hipStreamWaitValue64(sp, C.c_void_p(flag_mine), 5, 0, 0xFFFFFFFFFFFFFFFF), "StreamWaitValue64(own hipMalloc memory)")
        out = (C.c_ubyte * 16)()
        chk(hip, hip.hipStreamSynchronize(sp), "sync")
        chk(hip, hip.hipMemcpy(out, C.c_void_p(data_mine), 16, 2), "D2H")
        print("rank0 one-shot: wait rc", rc, "data", list(out)[:4], flush=True)
    else:
        time.sleep(0.5)
        chk(hip, hip.hipMemsetAsync(C.c_void_p(data_peer), 7, 4096, sp), "memset peer data")
        rc = chk(hip, hip.hipStreamWriteValue64(sp, C.c_void_p(flag_peer), 5, 0), "StreamWriteValue64(peer memory)")
        chk(hip, hip.hipStreamSynchronize(sp), "sync")
        print("rank1 one-shot: write rc", rc, flush=True)
    q_out.put("done1")
    q_in.get()
    # --- ping-pong: r -> wait own flag >= 10+r ; write peer flag = 10+r   (rank 1 starts by writing)
    t0 = time.perf_counter()
    for r in range(rounds):
        v = 10 + r
        if rank == 1:
            hip.hipStreamWriteValue64(sp, C.c_void_p(flag_peer), v, 0)
            hip.hipStreamWaitValue64(sp, C.c_void_p(flag_mine), v, 0, 0xFFFFFFFFFFFFFFFF)
        else:
            hip.hipStreamWaitValue64(sp, C.c_void_p(flag_mine), v, 0, 0xFFFFFFFFFFFFFFFF)
            hip.hipStreamWriteValue64(sp, C.c_void_p(flag_peer), v, 0)
    t_issue = time.perf_counter() - t0
    chk(hip, hip.hipStreamSynchronize(sp), "sync pingpong")
    dt = time.perf_counter() - t0
    print(f"rank{rank} ping-pong {rounds} rounds: issue {t_issue*1e6/rounds:.1f} us/round, complete {dt*1e6/rounds:.1f} us/round", flush=True)
    q_out.put("done2")
    q_in.get()


if __name__ == "__main__":
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    qa, qb = ctx.Queue(), ctx.Queue()
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    p0 = ctx.Process(target=worker, args=(0, qa, qb, rounds))
    p1 = ctx.Process(target=worker, args=(1, qb, qa, rounds))
    p0.start(); p1.start()
    p0.join(120); p1.join(120)
    print("exit codes", p0.exitcode, p1.exitcode)
