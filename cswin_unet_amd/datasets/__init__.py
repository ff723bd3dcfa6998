from .dataset_synapse import RandomGenerator, Synapse_dataset, random_rot_flip, random_rotate, write_synthetic_synapse  # noqa: F401
