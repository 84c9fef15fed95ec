/*
 * astar_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE: interface of oracle/astar_oracle.c, the CPU restatement of the
 * reference's kinodynamic A* (path_searching/src/kinodynamic_astar.cpp) and of the occupancy queries it makes
 * (occ_grid/src/occ_map.cpp, raycast.cpp).  PARITY UNPINNED (see the header of astar_oracle.c).
 * Only tests/ and tests/tools/ may load it.
 */
#ifndef ASTAR_ORACLE_H
#define ASTAR_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* KinodynamicAstar::search return values (kinodynamic_astar.h:169) */
#define ORC_ASTAR_REACH_HORIZON 1
#define ORC_ASTAR_REACH_END 2
#define ORC_ASTAR_NO_PATH 3
#define ORC_ASTAR_REACH_END_BUT_SHOT_FAILS 4

#define ORC_ASTAR_MAX_INPUTS 216
#define ORC_ASTAR_MAX_DURATIONS 16
#define ORC_ASTAR_MAX_PATH 256

typedef struct {
    /* occupancy map (OccMap): occ[x][y][z] != 0 <=> occupancy_buffer_ > min_occupancy_log_ (occ_map.cpp:105) */
    const unsigned char *occ;
    int grid[3];          /* grid_size_ = ceil(map_size_ / resolution_) (occ_map.cpp:789)                                  */
    double origin[3];     /* occ_map/origin_*                                                                              */
    double map_size[3];   /* occ_map/map_size_* (the search bounds states by map_size * 0.5, kinodynamic_astar.cpp:152-154) */
    double resolution;    /* occ_map/resolution = the A* voxel size (intialGridMap, kinodynamic_astar.cpp:511)              */
    int use_local;        /* 0: every voxel is "local"; 1: voxels outside [local_min, local_max] read as free (isInLocalMap) */
    int local_min[3], local_max[3];
    double ego_r, ego_h;  /* nmpc/ego_r, nmpc/ego_h (occ_map.cpp:764-765)                                                  */
    /* search parameters (setParam, kinodynamic_astar.cpp:290-305; launch values advanced_param.xml:97-109) */
    double max_tau, init_max_tau, max_vel, max_acc, w_time, horizon, lambda_heu;
    int allocate_num, check_num;
    double tie_breaker;   /* 1 + 1 / 10000 (kinodynamic_astar.h:139) */
    int max_expand;       /* capacity of tmp_expand_nodes (>= inputs x durations) */
} orc_astar_params;

typedef struct {
    int status, use_node_num, iter_num, is_shot_succ, n_path;
    double coef_shot[12]; /* coef_shot_(dim, power of t) */
    double t_shot;
    double path_state[ORC_ASTAR_MAX_PATH][6], path_input[ORC_ASTAR_MAX_PATH][3], path_duration[ORC_ASTAR_MAX_PATH];
    int path_node[ORC_ASTAR_MAX_PATH]; /* pool index of each path node = order of creation */
} orc_astar_result;

double orc_det_cbrt(double x);
double orc_det_acos(double x);
double orc_det_cos(double x);

int orc_astar_search(const orc_astar_params *P, const double start_pt[3], const double start_v[3], const double start_a[3],
                     const double end_pt[3], const double end_v[3], int init, const double external_acc[3], orc_astar_result *out);
int orc_astar_kino_traj(const orc_astar_params *P, const double external_acc[3], const orc_astar_result *r, double delta_t, double *pts, int cap);
int orc_astar_replay(const orc_astar_params *P, const double external_acc[3], const orc_astar_result *r);
int orc_astar_plan(const orc_astar_params *P, const double start_pt[3], const double start_v[3], const double start_a[3],
                   const double end_pt[3], const double end_v[3], int init, const double external_acc[3], double Ts,
                   double *kino_path, int cap, int *kino_size, orc_astar_result *res, int *retried, const double *retry_pt, const double *retry_v);
void orc_astar_batch(int B, const orc_astar_params *P, const double *start_pt, const double *start_v, const double *start_a,
                     const double *end_pt, const double *end_v, int init, const double *external_acc, double Ts, double *kino_path,
                     int cap, int *kino_size, int *status, orc_astar_result *res, int *retried, int nthreads, const double *retry_pt, const double *retry_v);
#ifdef __cplusplus
}
#endif
#endif
