// definitions of the castro:: runtime parameters (reference defaults, Source/driver/_cpp_parameters) and stub statics
#include <Castro.H>
namespace stub_eos { Real gamma = 1.4; }
namespace castro {
    Real small_dens = 1.e-100, small_pres = 1.e-100, small_temp = 1.e-100, small_ener = 1.e-100, T_guess = 1.e8, cg_tol = 1.0e-5,
         difmag = 0.1, cfl = 0.5, speed_limit = 0.0;
    Real dual_energy_eta1 = 1.0, dual_energy_eta2 = 1.0e-4, retry_small_density_cutoff = -1.e200, pslope_cutoff_density = -1.e20;
    int cg_maxiter = 12, cg_blend = 2, riemann_solver = 0, ppm_temp_fix = 0, hybrid_riemann = 0, use_reconstructed_gamma1 = 0;
    int transverse_use_eos = 0, transverse_reset_density = 1, transverse_reset_rhoe = 0, time_integration_method = 0;
    int source_term_predictor = 0, do_rotation = 0, state_in_rotating_frame = 1, ppm_type = 1, use_flattening = 1, first_order_hydro = 0;
    int plm_iorder = 2, plm_limiter = 2, use_pslope = 1, do_hydro = 1, verbose = 0, limit_fluxes_on_small_dens = 0, limit_fluxes_on_large_vel = 0;
    int allow_small_energy = 1, allow_negative_energy = 0, hybrid_hydro = 0, do_sponge = 0, density_reset_method = 1, do_grav = 0, do_react = 0;
    int mol_order = 2, do_ctu = 1, sdc_order = 2, domain_is_plane_parallel = 0;
    Real rotational_period = -1.e200, rotational_dPdt = 0.0;
    int rot_axis = 3, rot_source_type = 4, implicit_rotation_update = 1, rotation_include_centrifugal = 1, rotation_include_coriolis = 1;
}
Geometry Castro::geom;
BCRec Castro::phys_bc;
int Castro::verbose = 0;
int Castro::NUM_GROW = 4;
// the union of the Sedov and Sod problem parameters (stub/prob_parameters.H), reference defaults of the two _prob_params
namespace problem {
    Real center[3] = {0.0, 0.0, 0.0};
    Real p_ambient = 1.e-5, dens_ambient = 1.0, exp_energy = 1.0, temp_ambient = -1.e2, e_ambient = 0.0, r_init = 0.05, e_exp = 0.0;
    int nsub = 4;
    Real p_l = 1.0, u_l = 0.0, rho_l = 1.0, p_r = 0.1, u_r = 0.0, rho_r = 0.125, rhoe_l = 0.0, rhoe_r = 0.0, frac = 0.5, T_l = 1.0, T_r = 1.0,
         split[3] = {0.0, 0.0, 0.0};
    int use_Tinit = 0, idir = 1;
}
