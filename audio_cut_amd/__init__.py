"""audio-cut hot path (separate + framewise features + pause detection + cut refinement) on MI355X.

HIP kernels behind a C ABI (`include/audiocut_hip.h`, `csrc/`) driven by Python host code that keeps
the reference's plug points: `api.separate_and_segment`, `core.EnhancedVocalSeparator`,
`utils.gpu_pipeline`, `analysis.ChunkFeatureBuilder / TrackFeatureCache`,
`detectors.SileroChunkVAD / PureVocalPauseDetector`, `cutting.refine.finalize_cut_points`.
"""
__version__ = "0.1.0"
