"""Host-side mirror of the reference's batch format (manner/data/components) for the device-side collate."""
