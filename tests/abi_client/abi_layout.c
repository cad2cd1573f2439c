/* tests/abi_client/abi_layout.c -- TEST INFRASTRUCTURE.  Compiled as C11 (gcc) and as C++17 (g++) against include/c8.h:
 * the header itself is what is compiled, no device needed.  Prints sizeof / offsetof of every struct of the C ABI as JSON;
 * tests/test_abi.py compares both outputs with the ctypes mirror in calibr8_amd/lib.py (evaluations.hpp:23-84 is the
 * seam these structs stand at). */
#include <stddef.h>
#include <stdio.h>

#include "c8.h"

#define S(type) printf("%s \"%s\": {\"sizeof\": %zu", first++ ? ",\n" : "", #type, sizeof(type))
#define F(type, field) printf(", \"%s\": %zu", #field, offsetof(type, field))
#define E() printf("}")

int main(void) {
  int first = 0;
  printf("{\n");
  S(c8_mesh_desc); F(c8_mesh_desc, elem_type); F(c8_mesh_desc, num_nodes); F(c8_mesh_desc, num_elems); F(c8_mesh_desc, num_elem_sets);
  F(c8_mesh_desc, coords); F(c8_mesh_desc, conn); F(c8_mesh_desc, elem_set); F(c8_mesh_desc, num_extra_pairs); F(c8_mesh_desc, extra_pairs); E();
  S(c8_model_desc); F(c8_model_desc, global_type); F(c8_model_desc, local_type); F(c8_model_desc, stabilization_multiplier);
  F(c8_model_desc, local_max_iters); F(c8_model_desc, local_abs_tol); F(c8_model_desc, local_rel_tol); F(c8_model_desc, num_params);
  F(c8_model_desc, params); F(c8_model_desc, thickness); F(c8_model_desc, ls_sufficient_decrease); F(c8_model_desc, ls_min_backtrack);
  F(c8_model_desc, ls_max_backtrack); F(c8_model_desc, ls_max_evals); E();
  S(c8_state); F(c8_state, x); F(c8_state, x_prev); F(c8_state, xi_prev); F(c8_state, xi); E();
  S(c8_system); F(c8_system, A); F(c8_system, b); E();
  S(c8_calibration_desc); F(c8_calibration_desc, num_faces); F(c8_calibration_desc, nodes_per_face); F(c8_calibration_desc, faces);
  F(c8_calibration_desc, weights); F(c8_calibration_desc, balance_factor); F(c8_calibration_desc, coord_idx);
  F(c8_calibration_desc, coord_value); F(c8_calibration_desc, coord_tol); F(c8_calibration_desc, reaction_comp);
  F(c8_calibration_desc, dt_over_total_time); E();
  S(c8_dbc); F(c8_dbc, resid); F(c8_dbc, eq); F(c8_dbc, n); F(c8_dbc, nodes); F(c8_dbc, values); E();
  S(c8_tbc); F(c8_tbc, resid); F(c8_tbc, n); F(c8_tbc, nodes_per_face); F(c8_tbc, faces); F(c8_tbc, traction); E();
  S(c8_newton_opts); F(c8_newton_opts, max_iters); F(c8_newton_opts, abs_tol); F(c8_newton_opts, rel_tol); F(c8_newton_opts, line_search);
  F(c8_newton_opts, sufficient_decrease); F(c8_newton_opts, min_backtrack); F(c8_newton_opts, max_backtrack); F(c8_newton_opts, max_evals); E();
  S(c8_halo_desc); F(c8_halo_desc, num_owned); F(c8_halo_desc, num_touched); F(c8_halo_desc, send_ptr); F(c8_halo_desc, send_nodes);
  F(c8_halo_desc, recv_ptr); F(c8_halo_desc, recv_nodes); F(c8_halo_desc, recv_col_ptr); F(c8_halo_desc, recv_cols);
  F(c8_halo_desc, import_ptr); F(c8_halo_desc, import_nodes); F(c8_halo_desc, export_ptr); F(c8_halo_desc, export_nodes);
  F(c8_halo_desc, num_dims); F(c8_halo_desc, num_residuals); E();
  S(c8_lbfgs_opts); F(c8_lbfgs_opts, max_iters); F(c8_lbfgs_opts, grad_tol); F(c8_lbfgs_opts, step_tol); F(c8_lbfgs_opts, max_ls_evals);
  F(c8_lbfgs_opts, memory); E();
  S(c8_lbfgs_result); F(c8_lbfgs_result, iters); F(c8_lbfgs_result, evals); F(c8_lbfgs_result, status); F(c8_lbfgs_result, f);
  F(c8_lbfgs_result, projected_gradient_norm); E();
  printf(",\n \"enums\": {\"C8_OK\": %d, \"C8_LOCAL_SOLVE_FAILED\": %d, \"C8_ERR_ARG\": %d, \"C8_ERR_DEVICE\": %d, \"C8_ERR_UNSUPPORTED\": %d, "
         "\"C8_NOT_CONVERGED\": %d, \"C8_SCATTER_ATOMIC\": %d, \"C8_SCATTER_COLORED\": %d, \"C8_SCATTER_GATHER\": %d, \"C8_KERNEL_AUTO\": %d, "
         "\"C8_KERNEL_SLOT\": %d, \"C8_KERNEL_WAVE\": %d, \"C8_KERNEL_WAVE_AD\": %d, \"C8_KERNEL_NODE\": %d, \"C8_ELEM_TRI3\": %d, "
         "\"C8_ELEM_TET4\": %d, \"C8_ELEM_HEX8\": %d, \"C8_HALO_B\": %d, \"C8_HALO_A\": %d, \"C8_COMM_ID_BYTES\": %d, \"C8_SCALE_NONE\": %d, "
         "\"C8_SCALE_LOG\": %d, \"C8_SCALE_BOUNDS\": %d}\n}\n",
         C8_OK, C8_LOCAL_SOLVE_FAILED, C8_ERR_ARG, C8_ERR_DEVICE, C8_ERR_UNSUPPORTED, C8_NOT_CONVERGED, C8_SCATTER_ATOMIC, C8_SCATTER_COLORED,
         C8_SCATTER_GATHER, C8_KERNEL_AUTO, C8_KERNEL_SLOT, C8_KERNEL_WAVE, C8_KERNEL_WAVE_AD, C8_KERNEL_NODE, C8_ELEM_TRI3, C8_ELEM_TET4,
         C8_ELEM_HEX8, C8_HALO_B, C8_HALO_A, C8_COMM_ID_BYTES, C8_SCALE_NONE, C8_SCALE_LOG, C8_SCALE_BOUNDS);
  return 0;
}
