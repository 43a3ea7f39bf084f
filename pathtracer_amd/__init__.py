"""pathtracer_amd — MI355X-native radiance kernel behind the reference's host pipeline.

Layout:
  csrc/host/   C++ host pipeline (glTF, tangents, SAH BVH, light table, tonemap, PNG, flags)
  csrc/core/   hand-written HIP kernels for gfx950 + the C ABI of include/ptcore.h
  host.py      handles on the host pipeline objects (Scene, BVH, LST, write_image)
  core.py      handle on the device side (Renderer)
  render.py    batch loop / multi-GPU sample sharding (the role of src/main.cu)
  scenes.py    procedural stand-in scenes for configurations whose assets do not ship
"""
