"""MI355X-native drop-in for the ``sesameai`` package of zenoran/sesameai-tts: same module
names (``sesameai.models``, ``sesameai.generator``), hot path in gfx950 HIP kernels behind
libcsm_hip.so.  Put ``sesameai-tts_amd/`` on ``sys.path`` and the reference's
``from sesameai.generator import Segment, load_csm_1b`` (tts_service.py:22) resolves here."""
