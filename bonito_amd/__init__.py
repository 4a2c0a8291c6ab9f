"""
bonito_amd -- MI355X (gfx950) native engine for bonito's chunked-signal inference hot path
(signal chunks -> encoder -> CRF scores -> Viterbi / beam decode), behind the reference's own plugin
surface: ``config["model"]["package"]`` may name ``bonito_amd.crf`` / ``bonito_amd.transformer`` /
``bonito_amd.ctc`` and resolves to ``Model`` and ``basecall`` exactly as ``bonito.util.load_symbol``
expects (/root/reference bonito/util.py:223-234). Reference package names (``bonito.crf`` ...) found in
existing config.toml files are mapped onto these modules by ``bonito_amd.util.load_symbol``.

All device arithmetic lives in ``libbonito_hip.so`` (hand-written HIP; C ABI in include/bonito_hip.h).
"""
import os as _os

# The basecall pipeline drives one GPU from several HIP streams (H2D copy, encoder, decoder, the beam decoder's helper,
# torch's default stream). HIP multiplexes streams onto 4 hardware queues unless told otherwise, and streams that share a
# queue serialise; ask for 8 unless the user chose a value. Read by the runtime when it initialises, i.e. at the first HIP
# call of the process -- importing this package before touching the GPU is enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
