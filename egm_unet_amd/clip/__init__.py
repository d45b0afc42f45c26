from .clip import load, tokenize  # noqa: F401  (clip/__init__.py:1 exports the same names)
