"""Committed PyTorch-TunableOp selections for the library GEMMs of the dense layers (MI355X, ONCE-16k batch-2 step)."""
import os

TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_mi355x_once16k_b2.csv")


def enable_tuned_gemms(path=TUNING_FILE):
    """The dense layers run on hipBLASLt through torch; its default heuristics pick poor kernels for
    the backward GEMMs of this model (tall-skinny weight gradients with K = 65k-262k tokens).
    PyTorch's TunableOp benchmarks every GEMM shape once and records the fastest solution; the
    recorded choices for the ONCE-16k / batch-2 step (12 min of tuning on one MI355X,
    `tools/tune_gemms.sh`) are committed and only LOADED here (tuning stays off, unknown shapes use the
    default heuristic).  Measured: 78.2 -> 58.3 ms per training step."""
    try:
        import torch.cuda.tunable as tn
        if not os.path.exists(path):
            return False
        tn.enable(True)
        tn.tuning_enable(False)
        tn.record_untuned_enable(False) if hasattr(tn, "record_untuned_enable") else None
        return bool(tn.read_file(path))
    except Exception:  # noqa: BLE001  (TunableOp is an optimisation, never a requirement)
        return False
