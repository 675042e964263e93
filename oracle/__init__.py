"""ORACLE - test infrastructure only.

CPU (torch fp32) restatement of the reference's CosyVoice3 inference path, each
function citing the reference file:line it follows.  Pinned against outputs of
the reference itself (tests/golden/, minted by tests/golden/mint_goldens.py).
Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package fangyan_tts_amd never imports it.
"""
