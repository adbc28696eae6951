"""firewheel_amd — MI355X-native per-block DSP executor for Firewheel audio graphs.

The product is `csrc/libfwgpu.so` (hand-written gfx950 HIP kernels behind the C ABI in
`include/fwgpu.h`).  This package is the thin Python host side used by the tests and `bench.py`:
it mirrors the reference's edit/process surface (firewheel-graph `AudioGraph`, `FirewheelProcessor`,
the basic nodes) one-to-one on top of the C ABI via ctypes.  There is no CPU fallback: importing works
anywhere (so the CPU test tier can check the ABI), but creating a context without a gfx950 device raises.
"""
from ._lib import LIB_PATH, FwgpuError, build_library, load_library  # noqa: F401
from .graph import (  # noqa: F401
    AddEdgeError,
    BeepTestNode,
    BiquadNode,
    CompileGraphError,
    DelayNode,
    DummyAudioNode,
    FirReverbNode,
    FirewheelGpuCtx,
    HardClipNode,
    HostNode,
    LoopRange,
    MonoToStereoNode,
    ResamplerNode,
    SampleFormat,
    SamplerNode,
    SpatialNode,
    StereoPanNode,
    StereoToMonoNode,
    StereoWidthNode,
    SumNode,
    VolumeNode,
)

__all__ = [
    "FirewheelGpuCtx", "HostNode", "VolumeNode", "SumNode", "SamplerNode", "BeepTestNode", "HardClipNode", "MonoToStereoNode",
    "StereoToMonoNode", "DummyAudioNode", "StereoPanNode", "StereoWidthNode", "BiquadNode", "DelayNode", "FirReverbNode", "ResamplerNode", "SpatialNode", "LoopRange", "SampleFormat", "AddEdgeError",
    "CompileGraphError", "FwgpuError", "load_library", "build_library", "LIB_PATH",
]
