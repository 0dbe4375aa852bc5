import os, sys, torch
sys.path.insert(0, '.')
from benchmarks import workloads
def run(tail, chain):
    os.environ["PDA_GRAPH_TAIL"] = tail
    from pdanet_amd import pointnet2_utils as pu
    pu.SA_WIDE_CHAIN = chain
    torch.manual_seed(0)
    wl = workloads.create("detector_train", 2, 16384, torch.device("cuda:0"), 0, 1)
    losses = [float(wl.step().detach()) for _ in range(6)]
    print("graph_tail", tail, "wide_chain", chain, ["%.6f" % l for l in losses], flush=True)
    del wl
for tail, chain in (("0", True), ("0", True), ("1", True), ("1", True), ("0", False), ("0", False), ("1", False)):
    run(tail, chain)
