from .pointclouds import Pointclouds, join_pointclouds_as_batch

__all__ = ["Pointclouds", "join_pointclouds_as_batch"]
