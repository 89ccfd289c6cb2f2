from .pointclouds import (Pointclouds, all_close, get_bounding_boxes, join_pointclouds_as_batch,
                          join_pointclouds_as_scene, offset, scale, subsample)

__all__ = ["Pointclouds", "all_close", "get_bounding_boxes", "join_pointclouds_as_batch", "join_pointclouds_as_scene",
           "offset", "scale", "subsample"]
