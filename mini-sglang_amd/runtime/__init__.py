"""Caller side of the hot path: batch tensors, pools' users, a Qwen3-shaped module stack, model and graph runners."""
