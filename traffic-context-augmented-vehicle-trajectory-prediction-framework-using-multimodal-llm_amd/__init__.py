"""MI355X-native hot path of the multimodal-LLM trajectory predictor.

Import this package as ``tcavt_amd`` (see ``/tcavt_amd/__init__.py`` at the repo
root: the directory name required by the build contract is not an identifier).
"""
