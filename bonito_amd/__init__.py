"""
bonito_amd -- MI355X (gfx950) native engine for bonito's chunked-signal inference hot path
(signal chunks -> encoder -> CRF scores -> Viterbi / beam decode), behind the reference's own plugin
surface: ``config["model"]["package"]`` may name ``bonito_amd.crf`` / ``bonito_amd.transformer`` /
``bonito_amd.ctc`` and resolves to ``Model`` and ``basecall`` exactly as ``bonito.util.load_symbol``
expects (/root/reference bonito/util.py:223-234). Reference package names (``bonito.crf`` ...) found in
existing config.toml files are mapped onto these modules by ``bonito_amd.util.load_symbol``.

All device arithmetic lives in ``libbonito_hip.so`` (hand-written HIP; C ABI in include/bonito_hip.h).
"""
__version__ = "0.1.0"
