"""readtape_amd — MI355X-native analog front end for magnetic-tape waveform decoding.

Sub-modules:
  tbin      TBIN container reader/writer (host side)
  synth     synthetic tape generator (tests / benchmarks)
  frontend  the HIP front end behind the C ABI of include/rt_frontend.h (needs the built .so)
"""
__version__ = "0.1.0"
