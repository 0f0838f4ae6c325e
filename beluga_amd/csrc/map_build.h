// map_build.h — host-side, once-per-map work: the likelihood field and the free-cell list.
//
// This is NOT on the per-cycle path (likelihood_field_model_base.hpp:96-99 runs it in the sensor
// model's constructor / update_map).  The reference's distance map is not an exact Euclidean
// transform: a cell inherits the nearest obstacle of whichever 4-neighbour reaches it first out of
// a std::priority_queue (distance_map.hpp:74-95), so its values depend on that container's
// tie-breaking.  To hand beluga users the same field bit for bit, the build stays a host wavefront
// over the same container; a device EDT is listed under "next" (SURVEY.md §8f rank 1).
#pragma once
#include <cstdint>
#include <vector>

#include "beluga_mcl.h"

namespace mcl {

struct OccupancyTraits {
  int8_t free_value, unknown_value, occupied_value;
};

// likelihood_field_model_base.hpp:130-185
void build_likelihood_field(const int8_t* cells, uint32_t W, uint32_t H, double resolution, const OccupancyTraits& traits,
                            const mcl_lf_params& params, std::vector<float>& field);

// occupancy_grid.hpp:164-171 (free_cells): linear indices of free cells, ascending.
void collect_free_cells(const int8_t* cells, uint32_t W, uint32_t H, const OccupancyTraits& traits, std::vector<uint32_t>& out);

}  // namespace mcl
