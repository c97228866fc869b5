"""MI355X-native D2Q9 lid-driven-cavity lattice-Boltzmann hot path (drop-in for the
reference's MRT_GPU.py GPU path): Python host -> C ABI (include/lbm.h) -> HIP kernels."""
from .solver import CavityBatch, CavitySolver, relaxation, comm_unique_id, launch_plan  # noqa: F401

__all__ = ["CavityBatch", "CavitySolver", "relaxation", "comm_unique_id", "launch_plan"]
