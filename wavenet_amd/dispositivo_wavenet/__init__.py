"""Host-side mirror of the reference package `dispositivo_wavenet` (layer 1)."""
