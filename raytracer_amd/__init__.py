"""raytracer_amd — MI355X-native wavefront path tracer behind the host surface of Davvos11/raytracer.

Only the per-pixel ``ray_color`` path is here: ``HittableList`` (scene + acceleration structure), ``Camera.render``
and ``run``.  All compute is in ``lib/librt_hip.so`` (HIP, gfx950) behind ``include/rt_hip.h``.
"""
from .host import Camera, Comm, HittableList, RtError, init, run, set_profiling  # noqa: F401
