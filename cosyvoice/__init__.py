"""Import shim: lets `from cosyvoice.cli.cosyvoice import AutoModel` (compare_inference.py:31,
dialect_inference_test.py:30) resolve to the MI355X build when this repository is on sys.path."""
