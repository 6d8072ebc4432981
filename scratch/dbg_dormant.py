import sys, numpy as np
sys.path.insert(0,'/root/repo')
from audio_cut_amd import _native, config as PCFG
from audio_cut_amd.testing import signals
from audio_cut_amd.detectors.pure_vocal_pause_detector import PureVocalPauseDetector
hip=_native.Context()
g=np.load('/root/repo/tests/golden/dormant_branch.npz')
PCFG.set_runtime_config({"pure_vocal_detection.enable_relative_energy_mode": False})
for tag,x in (("voice",signals.voice_with_rests(14.0,seed=3)),("bursts",signals.c1_sine_silence(12.0,seed=2))):
    det=PureVocalPauseDetector(44100,ctx=hip)
    ft=det._extract_vocal_features(hip.to_device(x))
    f0r=g[tag+"_f0"]; vpr=g[tag+"_voiced_prob"]
    nanm=np.isnan(ft.f0_contour)!=np.isnan(f0r)
    print(tag,"nan mismatch frames:", np.flatnonzero(nanm)[:20], nanm.sum())
    v=~np.isnan(ft.f0_contour)&~np.isnan(f0r)
    bad=np.flatnonzero(v & (np.abs(ft.f0_contour-f0r)>1e-9*np.abs(f0r)))
    print(" f0 mismatches:", bad[:20], len(bad))
    print(" vp max abs diff", np.abs(ft.f0_confidence-vpr).max(), np.argmax(np.abs(ft.f0_confidence-vpr)))
    print(" rms", np.abs(ft.rms_energy-g[tag+"_rms"]).max(), "centroid rel", np.max(np.abs(ft.spectral_centroid-g[tag+"_centroid"])/(np.abs(g[tag+"_centroid"])+1e-2)),
          "ratio", np.abs(ft.harmonic_ratio-g[tag+"_harmonic_ratio"]).max(), "zcr", np.abs(ft.zero_crossing_rate-g[tag+"_zcr"]).max())
    for j in range(3):
        r=g[f"{tag}_formant{j}"]; a=ft.formant_energies[j]
        print(" formant",j,len(a),len(r), (np.max(np.abs(a[:min(len(a),len(r))]-r[:min(len(a),len(r))])/(np.abs(r[:min(len(a),len(r))])+1e-6)) if len(r) else None))
    print(" cand", det._detect_candidate_pauses(ft), g[tag+"_candidates"].tolist())
