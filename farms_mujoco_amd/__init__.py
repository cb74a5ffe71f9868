"""farms_mujoco_amd — MI355X-native batched step / drag / readout behind farms_mujoco's API shape."""
