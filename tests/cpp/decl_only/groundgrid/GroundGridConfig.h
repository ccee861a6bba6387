#pragma once
// declaration-only stand-in (see README.md) for the header dynamic_reconfigure generates from cfg/GroundGrid.cfg:8-21
// (int_t -> int, double_t -> double; field names as in the .cfg)
namespace groundgrid {
class GroundGridConfig {
  public:
    int point_count_cell_variance_threshold;
    int max_ring;
    double groundpatch_detection_minimum_threshold;
    double distance_factor;
    double minimum_distance_factor;
    double miminum_point_height_threshold;
    double minimum_point_height_obstacle_threshold;
    double outlier_tolerance;
    double ground_patch_detection_minimum_point_count_threshold;
    double patch_size_change_distance;
    double occupied_cells_decrease_factor;
    double occupied_cells_point_count_factor;
    double min_outlier_detection_ground_confidence;
    int thread_count;
};
}
