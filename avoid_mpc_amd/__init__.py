"""avoid_mpc_amd -- MI355X-native hot path of Avoid-MPC (dual KD-tree queries + MPC shooting solve).

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI of include/avoid_mpc_amd.h),
the ctypes binding of that ABI and the host-side mirror of the reference's interface.
"""
__version__ = "0.1.0"
