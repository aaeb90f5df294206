"""raytracer_amd — MI355X-native wavefront path-tracing hot path.

`host`        host scene model (C++ library, no GPU)
`pathtracer`  the device context: scene upload, state, render, ray-set queries
`ffi`         struct layouts and library loading
"""
from . import ffi, host  # noqa: F401
