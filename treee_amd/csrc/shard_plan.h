// shard_plan.h -- the plan object behind treee_shard_plan_* (shard_plan.cpp); used by the sharded build in octree.hip
#pragma once
#include "common.h"

struct treee_shard_plan {
	int world = 1;
	uint32_t L = 5, leaf_cap = 256, lod_cap = 256, ncell = 0;
	std::vector<uint32_t> counts;            // [world][ncell] points of every rank's input per level-L cell
	std::vector<uint64_t> prefix;            // [ncell + 1] exclusive prefix of the cell totals = global sorted position of a cell
	std::vector<uint64_t> index_base;        // [world + 1] global index of a rank's first input point
	std::vector<uint32_t> cell_begin;        // [world + 1] owned cells
	std::vector<treee_octree_node> top;      // levels < L, breadth first, global fields
	std::vector<treee_octree_node> level;    // level L, key order, global `first`
	std::vector<uint32_t> level_first;       // [world + 1] owned nodes of `level`
	uint32_t n_top_lod = 0;
	uint32_t top_begin[TREEE_SHARD_LEVELS] = {};
};
