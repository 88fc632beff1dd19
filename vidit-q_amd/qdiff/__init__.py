"""Host-side mirror of the reference's ``qdiff`` operator API."""
