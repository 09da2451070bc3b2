"""MI355X-native quantized-linear backend (bitsandbytes-style API)."""
