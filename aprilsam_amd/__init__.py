"""aprilsam_amd — MI355X-native replacement for AprilSAM's april_graph_cholesky / _inc hot path.

The product is the C-ABI shared library aprilsam_amd/lib/libaprilsam_amd.so (HIP kernels for gfx950 +
host runtime, sources under aprilsam_amd/csrc/, interface in include/aprilsam_amd.h).  This Python
package is only its host-side mirror for tests and benchmarks: `host` (ctypes binding with the
reference's API names), `abi` (struct layouts), `datasets` (M3500 / lattice inputs), `harness` (the
reference's two example drivers restated).
"""
__all__ = ["abi", "host", "datasets", "harness"]
