"""MI355X-native two-stream action-recognition inference path (Sheet03 of arindamrc/video_analytics).

Host-side mirror of the reference's call surface (``spatialModel`` / ``temporalModel`` /
``combinedModel`` / ``utils`` / ``parameters``) over hand-written gfx950 HIP kernels reached through
the C ABI of ``include/va.h`` (``libva_hip.so``).  See DESIGN.md.
"""
__version__ = "0.1.0"
