from fangyan_tts_amd.cli.cosyvoice import AutoModel, CosyVoice3  # noqa: F401
