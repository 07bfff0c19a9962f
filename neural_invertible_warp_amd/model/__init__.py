"""Model modules named as the reference's (`model/<name>.py` exposing Graph / NeRF), so that an
engine that locates them by name (reference train.py:23, model/base.py:35) finds the same classes."""
