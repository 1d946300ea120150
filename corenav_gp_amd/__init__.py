"""corenav_gp_amd: MI355X-native engine for the corenav-GP slip-GP hot path.

Host side mirrors the reference's own surface for this path:
  * gp_slip_node  -- drop-in for core_navigation/script/gp_slip_node.py (callback -> GP_Output)
  * GpPredictor   -- C++ class (csrc/gp_predictor.*), reached from Python through the C ABI
  * engine        -- ctypes binding of include/corenav_gp.h (libcorenav_gp.so, HIP/gfx950)
There is no CPU fallback: importing `engine` without the built HIP library raises.
"""
__version__ = "0.1.0"
