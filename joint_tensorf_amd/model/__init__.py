"""Engine modules exposing the reference's `model/<name>.py` interface (Model / Graph / NeRF)."""
