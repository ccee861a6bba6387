#pragma once
// declaration-only stand-in (see README.md)
namespace grid_map { class GridMapCvConverter; }
