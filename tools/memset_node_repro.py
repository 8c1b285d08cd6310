"""Stand-alone probe (torch + the HIP runtime only, nothing of this library): is a MEMSET NODE inside a captured hipGraph ordered
between the kernel nodes around it?  Round 5 found NeurComm's captured update reading flag words its memset node had not
cleared yet (profiles/r05_determinism.txt).  The graph below is that situation in miniature, per replay:
    K1  dirty  <- flags + 1 (flags become non-zero: "the previous replay's step counts"), plus some real work in front
    M   hipMemsetAsync(flags, 0)                         -> a memset node
    K2  seen   <- flags (what the consumer kernel reads)  -> must be all zero if M ran between K1 and K2
    python tools/memset_node_repro.py [replays] [words,words,...] [work_elems]
Prints how many words were ever read non-zero, for the memset-node form and for the same graph with the zeroing as a kernel."""
import ctypes as C
import sys

import torch

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
sizes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1 << 12, 1 << 16, 1 << 18, 1 << 20, 1 << 22]
work = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 22
hip = None
for line in open('/proc/self/maps'):
    if 'libamdhip64.so' in line:
        hip = C.CDLL(line.split()[-1])
        break
dev = torch.device('cuda')


def run(memset_node, words):
    flags = torch.zeros(words, dtype=torch.int32, device=dev)
    seen = torch.zeros(words, dtype=torch.int32, device=dev)
    bad = torch.zeros(words, dtype=torch.int32, device=dev)     # per word: in how many replays it was read non-zero
    a, b = torch.randn(work, device=dev), torch.randn(work, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        torch.add(a, b, out=a)                                # some work in front (a kernel node)
        flags.add_(1)                                         # K1
        if memset_node:
            hip.hipMemsetAsync(C.c_void_p(flags.data_ptr()), 0, C.c_size_t(words * 4), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        else:
            flags.mul_(0)                                     # the same zeroing as a kernel node
        torch.add(flags, 0, out=seen)                         # K2
        bad.add_((seen != 0).to(torch.int32))                 # (element-wise kernels only: an aten multi-block reduction would
                                                              #  bring memset nodes of its own -- its semaphores -- into the graph)
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    return int((bad != 0).sum().item()), int(bad.max().item())


# (flag buffers of 16 KB ... 16 MB: a few blocks of one XCD up to many blocks on all eight, whose L2s are not coherent with each other --
# the fill kernel behind a memset node need not run on the XCD whose L2 holds K1's lines)
for words in sizes:
    for form in (True, False):
        n_words, n_replays = run(form, words)
        print('%8d words, %-11s: %d words were read non-zero behind the zeroing (the worst one in %d of %d replays)'
              % (words, 'memset node' if form else 'kernel node', n_words, n_replays, replays))
