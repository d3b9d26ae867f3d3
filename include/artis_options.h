/*
 * artis_options.h -- compile-time options of the packet path.
 *
 * Mirrors the subset of the reference's artisoptions_*.h presets that the
 * r-packet / k-packet / macro-atom path reads. The defaults are
 * artisoptions_classic.h (the preset BASELINE.json's configs[0..2] name).
 * Like the reference, a different preset means a rebuild: pass -DARTIS_OPT_...
 * Names are the reference's option names with an ARTIS_OPT_ prefix.
 */
#ifndef ARTIS_OPTIONS_H
#define ARTIS_OPTIONS_H

/* PARTICLE_THERMALISATION_SCHEME (artisoptions_classic.h:146), numbered like enum class ParticleThermalisationScheme
 * (constants.h:82-89) */
#define ARTIS_PARTICLE_INSTANTFULLDEPOSITION 0
#define ARTIS_PARTICLE_TIMEDEPENDENT 1
#define ARTIS_PARTICLE_TIMEDEPENDENT_WITH_ADIABATIC_LOSS 2
#define ARTIS_PARTICLE_TIMEDEPENDENTWITHGAMMAPRODUCTS 3
#define ARTIS_PARTICLE_BARNES 4
#define ARTIS_PARTICLE_WOLLAEGER 5

/* ---- The option sets the reference's own CI runs (tests/setup_*.sh: an options file + sed edits). Each is pinned option by
 * option against the file the script makes (oracle/Makefile target `ref`: the script's own sed lines applied to a scratch
 * copy under oracle/_ref/, compiled with oracle/ref_harness/ref_options_main.cc; tests/golden/options_reference.json).
 *   -DARTIS_PRESET_CI_KILONOVA             setup_kilonova_1d.sh, setup_kilonova_2d.sh: kilonova_lte, TABLESIZE 20, 1000-20000 K
 *   -DARTIS_PRESET_CI_KILONOVA_BARNES      setup_kilonova_2d_barnesthermalisation.sh: + PARTICLE and GAMMA thermalisation BARNES
 *   -DARTIS_PRESET_CI_KILONOVA_EXPOPAC     setup_kilonova_2d_expansionopac.sh: + RPKT_USE_EXPANSION_OPACITIES,
 *                                          RPKT_BOUNDBOUND_THERMALISATION_PROBABILITY = 1. (rpkt.cc:626 draws no random number then)
 *   -DARTIS_PRESET_CI_KILONOVA_XCOM        setup_kilonova_2d_xcomgammaphotoion.sh: + USE_XCOM_GAMMAPHOTOION (element number
 *                                          densities from the per-cell mean atomic weights: artis_cellstate.elem_meanweight)
 *   -DARTIS_PRESET_CI_NEBULAR              setup_nebular_1d_3dgrid.sh: nltenebular, TABLESIZE 20, 2000-10000 K,
 *                                          FIRST_NLTE_RADFIELD_TIMESTEP 7
 *   -DARTIS_PRESET_CI_NEBULAR_LIMITBFEST   setup_nebular_1d_3dgrid_limitbfest.sh: + LEVEL_HAS_BFEST for the NLTE levels only
 *   -DARTIS_PRESET_CI_NLTEPHOTOSPHERIC     setup_nltephotospheric_dynamic_ion_range_1d_1dgrid.sh: TABLESIZE 40,
 *                                          FIRST_NLTE_RADFIELD_TIMESTEP 4, RADFIELDBINCOUNT 24
 *   -DARTIS_PRESET_CI_CLASSIC_VPKT         setup_classicmode_1d_3dgrid.sh: classic + VPKT_ON (virtual packets, vpkt.cc: at every
 *                                          emission and electron scattering a packet is traced towards each observer and
 *                                          its attenuated energy added to that observer's spectrum)
 *   -DARTIS_PRESET_CI_CLASSIC_VPKT_EXPOPAC setup_classicmode_3d.sh: + VPKT_USE_EXPANSION_OPACITIES (the virtual packets
 *                                          walk the binned expansion opacities beyond their first wavelength bin) */
#ifdef ARTIS_PRESET_CI_CLASSIC_VPKT
#define ARTIS_OPT_VPKT_ON 1
#endif
#ifdef ARTIS_PRESET_CI_CLASSIC_VPKT_EXPOPAC
#define ARTIS_OPT_VPKT_ON 1
#define ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES 1
#endif
#if defined(ARTIS_PRESET_CI_KILONOVA) || defined(ARTIS_PRESET_CI_KILONOVA_BARNES) || defined(ARTIS_PRESET_CI_KILONOVA_EXPOPAC) || \
    defined(ARTIS_PRESET_CI_KILONOVA_XCOM)
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_TABLESIZE 20
#define ARTIS_OPT_MINTEMP 1000.
#define ARTIS_OPT_MAXTEMP 20000.
#endif
#ifdef ARTIS_PRESET_CI_KILONOVA_BARNES
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_BARNES
#define ARTIS_OPT_GAMMA_THERMALISATION_SCHEME 1
#endif
#ifdef ARTIS_PRESET_CI_KILONOVA_EXPOPAC
#define ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES 1
#define ARTIS_OPT_RPKT_BB_THERMALISATION 1
#define ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY 1.f
#endif
#ifdef ARTIS_PRESET_CI_KILONOVA_XCOM
#define ARTIS_OPT_USE_XCOM_GAMMAPHOTOION 1
#endif
#if defined(ARTIS_PRESET_CI_NEBULAR) || defined(ARTIS_PRESET_CI_NEBULAR_LIMITBFEST)
#define ARTIS_PRESET_NLTENEBULAR
#define ARTIS_OPT_TABLESIZE 20
#define ARTIS_OPT_MINTEMP 2000.
#define ARTIS_OPT_MAXTEMP 10000.
#define ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP 7
#endif
#ifdef ARTIS_PRESET_CI_NEBULAR_LIMITBFEST
#define ARTIS_OPT_BFEST_SUBSET 1
#endif
#ifdef ARTIS_PRESET_CI_NLTEPHOTOSPHERIC
#define ARTIS_PRESET_NLTEPHOTOSPHERIC
#define ARTIS_OPT_TABLESIZE 40
#define ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP 4
#define ARTIS_OPT_RADFIELDBINCOUNT 24
#endif

/* -DARTIS_PRESET_KILONOVA_BARNES / _WOLLAEGER: artisoptions_kilonova_lte.h with the analytic thermalisation efficiency of
 * Barnes et al. (2016) or Wollaeger et al. (2018) instead of the local time-dependent scheme (update_packets.cc:69-88).
 * No options file of the reference selects them; built so that every branch of do_nonthermal_predeposit() is covered. */
#ifdef ARTIS_PRESET_KILONOVA_BARNES
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_BARNES
#endif
#ifdef ARTIS_PRESET_KILONOVA_WOLLAEGER
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_WOLLAEGER
#endif
/* -DARTIS_PRESET_KILONOVA_GAMMA_BARNES / _WOLLAEGER / _GUTTMAN: artisoptions_kilonova_lte.h with one of the parameterised
 * gamma-ray thermalisation schemes instead of gamma-ray transport (gammapkt.cc:775-866): a gamma packet is absorbed where
 * it is born with the scheme's deposition probability, or escapes. No options file of the reference selects them. */
#ifdef ARTIS_PRESET_KILONOVA_GAMMA_BARNES
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_GAMMA_THERMALISATION_SCHEME 1
#endif
#ifdef ARTIS_PRESET_KILONOVA_GAMMA_WOLLAEGER
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_GAMMA_THERMALISATION_SCHEME 2
#endif
#ifdef ARTIS_PRESET_KILONOVA_GAMMA_GUTTMAN
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_GAMMA_THERMALISATION_SCHEME 3
#endif
/* -DARTIS_PRESET_KILONOVA_GAMMA_GREY: gamma-ray transport with one grey absorption opacity (GAMMA_USE_KAPPA_GREY = 0.06 cm^2/g:
 * no Compton scattering, no pair production, gammapkt.cc:266, :420, :517, :553).
 * -DARTIS_PRESET_CLASSIC_GAMMA_XCOM: artisoptions_classic.h with USE_XCOM_GAMMAPHOTOION: the photoelectric opacity from the
 * tabulated XCOM cross sections of every element, log-log interpolated (gammapkt.cc:443-495; artis_model.xcom_*). */
#ifdef ARTIS_PRESET_KILONOVA_GAMMA_GREY
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_GAMMA_USE_KAPPA_GREY 1
#define ARTIS_OPT_GAMMA_KAPPA_GREY 0.06
#endif
#ifdef ARTIS_PRESET_CLASSIC_GAMMA_XCOM
#define ARTIS_OPT_USE_XCOM_GAMMAPHOTOION 1
#endif
/* -DARTIS_PRESET_KILONOVA_GAMMAPRODUCTS: TIMEDEPENDENTWITHGAMMAPRODUCTS (constants.h:86): a gamma-ray interaction hands its
 * energy to an electron / positron that thermalises with the local time-dependent scheme instead of depositing at once
 * (gammapkt.cc:404, :572, :630, :734, :925; update_packets.cc:174) */
#ifdef ARTIS_PRESET_KILONOVA_GAMMAPRODUCTS
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_TIMEDEPENDENTWITHGAMMAPRODUCTS
#endif

/* RPKT_USE_EXPANSION_OPACITIES = true (artisoptions_*.h:136; binned expansion opacities instead of the line-by-line walk,
 * rpkt.cc:221), which every options file of the reference leaves off, in its two forms:
 * -DARTIS_PRESET_KILONOVA_EXPOPAC: artisoptions_kilonova_lte.h + expansion opacities; the bin in which the event falls is
 *   re-traced line by line and a bound-bound event activates a macro-atom (relativistic Doppler branch of the bin walk);
 * -DARTIS_PRESET_CLASSIC_EXPOPAC_THERM: artisoptions_classic.h + expansion opacities +
 *   RPKT_BOUNDBOUND_THERMALISATION_PROBABILITY (:140; here 0.9): a bound-bound event thermalises or scatters instead of
 *   activating a macro-atom (rpkt.cc:624-648), k-packets in grey-free cells emit from kappa * B_nu (kpkt.cc:402). */
#ifdef ARTIS_PRESET_KILONOVA_EXPOPAC
#define ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES 1
#endif
#ifdef ARTIS_PRESET_CLASSIC_EXPOPAC_THERM
#define ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES 1
#define ARTIS_OPT_RPKT_BB_THERMALISATION 1
#define ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY 0.9f
#endif

/* -DARTIS_PRESET_KILONOVA_LTE: the packet-path options of artisoptions_kilonova_lte.h (BASELINE.json configs[3]);
 * every value below is the one of that file where it differs from artisoptions_classic.h. */
#ifdef ARTIS_PRESET_KILONOVA_LTE
#define ARTIS_OPT_DIPOLE 0                            /* artisoptions_kilonova_lte.h:48 */
#define ARTIS_OPT_POL_ON 0                            /* :49 */
#define ARTIS_OPT_MINPOP 1e-40                        /* :54 */
#define ARTIS_OPT_NU_MIN_R 1e13                       /* :56 */
#define ARTIS_OPT_NU_MAX_R 5e16                       /* :57 */
#define ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION 0    /* :59 */
#define ARTIS_OPT_DIRECT_COL_HEAT 1                   /* :37 */
#ifndef ARTIS_OPT_TABLESIZE
#define ARTIS_OPT_TABLESIZE 200                       /* :42 */
#define ARTIS_OPT_MINTEMP 500.                        /* :43 */
#define ARTIS_OPT_MAXTEMP 150000.                     /* :44 */
#endif
#define ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT 1    /* :118 */
#define ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT 1   /* :120 (on the packet path only the XCOM opacities and the NT_ON channels
                                                       * read element number densities: artis_cellstate.elem_meanweight) */
#ifndef ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_TIMEDEPENDENT /* :146 */
#endif
#endif

/* -DARTIS_PRESET_NLTENEBULAR: the packet-path options of artisoptions_nltenebular.h (BASELINE.json configs[4]): level
 * populations and photoionisation coefficients come from the host (NLTE solver, USE_LUT_PHOTOION off), the radiation
 * field is the multibin model (estimators per frequency bin, binned J_nu in the radiative excitation rates), every
 * continuum has a detailed bound-free estimator, and the non-thermal channels are on (NT_ON with a Spencer-Fano solution
 * from the host: ionisation / excitation branches of do_ntlepton_deposit(), non-thermal macro-atom rates). */
/* The reference's other three options files differ from artisoptions_nltenebular.h, on the packet path, in the values
 * below only (tests/golden/options_reference.json pins all six files):
 * -DARTIS_PRESET_CHRISTINENONTHERMAL   artisoptions_christinenonthermal.h
 * -DARTIS_PRESET_NLTEPHOTOSPHERIC      artisoptions_nltephotospheric_dynamic_ion_range.h (bound-free estimators for the NLTE
 *                                      levels only: LEVEL_HAS_BFEST :80, artis_model.allcont_bfestimindex)
 * -DARTIS_PRESET_NLTEWITHOUTNONTHERMAL artisoptions_nltewithoutnonthermal.h */
/* -DARTIS_PRESET_NLTENEBULAR_LINEEST: artisoptions_nltenebular.h + DETAILED_LINE_ESTIMATORS_ON (:78, off in every options
 * file of the reference): selected lines get their own intensity estimator, updated by every packet that redshifts
 * through them (radfield.cc:773, rpkt.cc:173-207) and used instead of the binned field in the radiative excitation rate
 * (macroatom.cc:628). artis_model.detailed_lineindices, artis_cellstate.Jb_lu_normed, artis_estimators.Jb_lu_*. */
#ifdef ARTIS_PRESET_NLTENEBULAR_LINEEST
#define ARTIS_PRESET_NLTENEBULAR
#define ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON 1
#endif
#ifdef ARTIS_PRESET_CHRISTINENONTHERMAL
#define ARTIS_PRESET_NLTENEBULAR
#define ARTIS_OPT_MINTEMP 3000.                       /* artisoptions_christinenonthermal.h:48 */
#define ARTIS_OPT_MAXTEMP 140000.                     /* :49 */
#define ARTIS_OPT_NU_MAX_R 5e16                       /* :62 */
#define ARTIS_OPT_RADFIELDBINCOUNT 64                 /* :68 */
#define ARTIS_OPT_RADFIELDBINS_NU_MAX (2.99792458e+10 / 500e-8)            /* :73 */
#define ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX (2.99792458e+10 / 50e-8) /* :74 */
#define ARTIS_OPT_NT_EXCITATION_ON 0                  /* :113 */
#endif
#ifdef ARTIS_PRESET_NLTEPHOTOSPHERIC
#define ARTIS_PRESET_NLTENEBULAR
#ifndef ARTIS_OPT_TABLESIZE
#define ARTIS_OPT_TABLESIZE 100                       /* artisoptions_nltephotospheric_dynamic_ion_range.h:45 */
#endif
#define ARTIS_OPT_MINTEMP 3500.                       /* :46 */
#define ARTIS_OPT_MAXTEMP 140000.                     /* :47 */
#define ARTIS_OPT_BFEST_SUBSET 1                      /* :80 LEVEL_HAS_BFEST: only some continua have an estimator */
#endif
#ifdef ARTIS_PRESET_NLTEWITHOUTNONTHERMAL
#define ARTIS_PRESET_NLTENEBULAR
#define ARTIS_OPT_DIPOLE 1                            /* artisoptions_nltewithoutnonthermal.h:51 */
#define ARTIS_OPT_POL_ON 1                            /* :52 */
#define ARTIS_OPT_NU_MIN_R 1e14                       /* :59 */
#define ARTIS_OPT_NU_MAX_R 5e16                       /* :60 */
#define ARTIS_OPT_TABLESIZE 200                       /* :45 */
#define ARTIS_OPT_MINTEMP 4000.                       /* :46 */
#define ARTIS_OPT_MAXTEMP 140000.                     /* :47 */
#define ARTIS_OPT_RADFIELDBINCOUNT 512                /* :66 */
#define ARTIS_OPT_RADFIELDBINS_NU_MAX (2.99792458e+10 / 100e-8)            /* :71 */
#define ARTIS_OPT_NT_EXCITATION_ON 0                  /* :111 */
#define ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP 1    /* :137 */
#endif
#ifdef ARTIS_PRESET_NLTENEBULAR
#ifndef ARTIS_OPT_DIPOLE
#define ARTIS_OPT_DIPOLE 0                            /* artisoptions_nltenebular.h:48 */
#define ARTIS_OPT_POL_ON 0                            /* :49 */
#endif
#define ARTIS_OPT_MINPOP 1e-40                        /* :54 */
#ifndef ARTIS_OPT_NU_MIN_R
#define ARTIS_OPT_NU_MIN_R 1e13                       /* :56 */
#endif
#define ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION 0    /* :59 */
#define ARTIS_OPT_DIRECT_COL_HEAT 1                   /* :36 */
#ifndef ARTIS_OPT_MINTEMP
#define ARTIS_OPT_MINTEMP 1000.                       /* :42 */
#define ARTIS_OPT_MAXTEMP 30000.                      /* :43 */
#endif
#define ARTIS_OPT_LTEPOP_EXCITATION_USE_TJ 0          /* :24 */
#define ARTIS_OPT_USE_LUT_PHOTOION 0                  /* :84 */
#define ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS 0      /* :86 */
#define ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON 1        /* :61 */
#define ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON 1         /* :78 (LEVEL_HAS_BFEST is true for every level, :80) */
#define ARTIS_OPT_NT_ON 1                             /* :102 with NT_SOLVE_SPENCERFANO :104 */
#endif
/* non-thermal options (artisoptions_nltenebular.h:110-119; NT_EXCITATION_ON only has a meaning with NT_ON) */
#ifndef ARTIS_OPT_NT_EXCITATION_ON
#define ARTIS_OPT_NT_EXCITATION_ON 1
#endif
#define ARTIS_OPT_NTEXCITATION_MAXNLEVELS_LOWER 5
#define ARTIS_OPT_NTEXCITATION_MAXNLEVELS_UPPER 250
#define ARTIS_OPT_NT_MAX_AUGER_ELECTRONS 2
/* multibin radiation field model (artisoptions_nltenebular.h:62-70) */
#ifndef ARTIS_OPT_RADFIELDBINCOUNT
#define ARTIS_OPT_RADFIELDBINCOUNT 256
#endif
#ifndef ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP
#define ARTIS_OPT_FIRST_NLTE_RADFIELD_TIMESTEP 12
#endif
#define ARTIS_OPT_RADFIELDBINS_NU_MIN (2.99792458e+10 / 40000e-8)
#ifndef ARTIS_OPT_RADFIELDBINS_NU_MAX
#define ARTIS_OPT_RADFIELDBINS_NU_MAX (2.99792458e+10 / 1085e-8)
#endif
#ifndef ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX
#define ARTIS_OPT_RADFIELDBINS_T_E_SUPERBIN_NU_MAX (2.99792458e+10 / 10e-8)
#endif
#ifndef ARTIS_OPT_BFEST_SUBSET
#define ARTIS_OPT_BFEST_SUBSET 0 /* LEVEL_HAS_BFEST is true for every level (nltenebular.h:82) or the estimators are off */
#endif

#ifndef ARTIS_OPT_DIPOLE
#define ARTIS_OPT_DIPOLE 1 /* artisoptions_classic.h:47 */
#endif
#ifndef ARTIS_OPT_POL_ON
#define ARTIS_OPT_POL_ON 1 /* artisoptions_classic.h:48 */
#endif
#ifndef ARTIS_OPT_MINPOP
#define ARTIS_OPT_MINPOP 1e-30 /* artisoptions_classic.h:53 */
#endif
#ifndef ARTIS_OPT_NU_MIN_R
#define ARTIS_OPT_NU_MIN_R 1e14 /* artisoptions_classic.h:55 */
#endif
#ifndef ARTIS_OPT_NU_MAX_R
#define ARTIS_OPT_NU_MAX_R 5e15 /* artisoptions_classic.h:56 */
#endif
#ifndef ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION
#define ARTIS_OPT_PHIXS_CLASSIC_NO_INTERPOLATION 1 /* artisoptions_classic.h:58 */
#endif
#ifndef ARTIS_OPT_USE_LUT_PHOTOION
#define ARTIS_OPT_USE_LUT_PHOTOION 1 /* artisoptions_classic.h:82 */
#endif
#ifndef ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS
#define ARTIS_OPT_USE_ION_BFHEATING_ESTIMATORS 1 /* artisoptions_classic.h:84 */
#endif
#ifndef ARTIS_OPT_LTEPOP_EXCITATION_USE_TJ
#define ARTIS_OPT_LTEPOP_EXCITATION_USE_TJ 1 /* artisoptions_classic.h:24 */
#endif
#ifndef ARTIS_OPT_DIRECT_COL_HEAT
#define ARTIS_OPT_DIRECT_COL_HEAT 0 /* artisoptions_classic.h:35 */
#endif
#ifndef ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP
#define ARTIS_OPT_BFCOOLING_USELEVELPOPNOTIONPOP 0 /* artisoptions_classic.h:135 */
#endif
#ifndef ARTIS_OPT_TABLESIZE
#define ARTIS_OPT_TABLESIZE 100 /* artisoptions_classic.h:40 */
#endif
#ifndef ARTIS_OPT_MINTEMP
#define ARTIS_OPT_MINTEMP 3500. /* artisoptions_classic.h:41 */
#endif
#ifndef ARTIS_OPT_MAXTEMP
#define ARTIS_OPT_MAXTEMP 140000. /* artisoptions_classic.h:42 */
#endif

#ifndef ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT
#define ARTIS_OPT_USE_RELATIVISTIC_DOPPLER_SHIFT 0 /* artisoptions_classic.h:118 */
#endif
#ifndef ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME
#define ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME ARTIS_PARTICLE_INSTANTFULLDEPOSITION
#endif
#define ARTIS_GAMMAPRODUCTS (ARTIS_OPT_PARTICLE_THERMALISATION_SCHEME == ARTIS_PARTICLE_TIMEDEPENDENTWITHGAMMAPRODUCTS)
#ifndef ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT
#define ARTIS_OPT_USE_CALCULATED_MEANATOMICWEIGHT 0 /* artisoptions_classic.h:120 */
#endif

/* Options of the reference this build does not implement: they must keep the
 * classic values. (A build that needs them fails here, not at run time.) */
#ifndef ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
#define ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES 0    /* artisoptions_classic.h:136 */
#endif
#ifndef ARTIS_OPT_RPKT_BB_THERMALISATION
#define ARTIS_OPT_RPKT_BB_THERMALISATION 0          /* RPKT_BOUNDBOUND_THERMALISATION_PROBABILITY.has_value(), :140 */
#define ARTIS_OPT_RPKT_BB_THERMALISATION_PROBABILITY 0.f
#endif
/* wavelength grid of the expansion opacities, rpkt.h:23-26 */
#define ARTIS_EXPOPAC_LAMBDAMIN 60.
#define ARTIS_EXPOPAC_LAMBDAMAX 40000.
#define ARTIS_EXPOPAC_DELTALAMBDA 20.
#define ARTIS_EXPOPAC_NBINS 1997 /* (lambdamax - lambdamin) / deltalambda */
#ifndef ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON
#define ARTIS_OPT_DETAILED_LINE_ESTIMATORS_ON 0     /* artisoptions_classic.h:71 */
#endif
#ifndef ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON
#define ARTIS_OPT_DETAILED_BF_ESTIMATORS_ON 0       /* artisoptions_classic.h:76 */
#endif
#ifndef ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON
#define ARTIS_OPT_MULTIBIN_RADFIELD_MODEL_ON 0      /* artisoptions_classic.h:60 */
#endif
#ifndef ARTIS_OPT_NT_ON
#define ARTIS_OPT_NT_ON 0                           /* artisoptions_classic.h:100 */
#endif
#ifndef ARTIS_OPT_VPKT_ON
#define ARTIS_OPT_VPKT_ON 0                         /* artisoptions_classic.h:49 */
#endif
#ifndef ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES
#define ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES 0    /* artisoptions_classic.h:138 */
#endif
#if ARTIS_OPT_VPKT_ON && !ARTIS_OPT_POL_ON
#error "VPKT_ON needs POL_ON (vpkt.cc:44)"
#endif
#if ARTIS_OPT_VPKT_ON && ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES
#error "VPKT cannot be used with r-packet expansion opacities (rpkt.cc:44)"
#endif
/* the per-cell tables of calculate_expansion_opacities() exist (rpkt.cc:953, :1055, :1102) */
#define ARTIS_EXPOPAC_TABLES (ARTIS_OPT_RPKT_USE_EXPANSION_OPACITIES || ARTIS_OPT_RPKT_BB_THERMALISATION || ARTIS_OPT_VPKT_USE_EXPANSION_OPACITIES)
/* gamma packets: the classic choices (artisoptions_classic.h:144-150) are the ones built */
#ifndef ARTIS_OPT_USE_XCOM_GAMMAPHOTOION
#define ARTIS_OPT_USE_XCOM_GAMMAPHOTOION 0            /* artisoptions_classic.h:144: Veigele fit for the photoelectric opacity */
#endif
#ifndef ARTIS_OPT_GAMMA_USE_KAPPA_GREY
#define ARTIS_OPT_GAMMA_USE_KAPPA_GREY 0             /* :150 std::nullopt: frequency-dependent opacities */
#define ARTIS_OPT_GAMMA_KAPPA_GREY 0.
#endif
/* GAMMA_THERMALISATION_SCHEME (artisoptions_classic.h:148), numbered like enum class GammaThermalisationScheme
 * (constants.h:81): 0 = FREQUENCYDEPENDENT (transport), 1 = BARNES, 2 = WOLLAEGER, 3 = GUTTMAN (gammapkt.cc:775-866) */
#define ARTIS_GAMMA_FREQUENCYDEPENDENT 0
#define ARTIS_GAMMA_BARNES 1
#define ARTIS_GAMMA_WOLLAEGER 2
#define ARTIS_GAMMA_GUTTMAN 3
#ifndef ARTIS_OPT_GAMMA_THERMALISATION_SCHEME
#define ARTIS_OPT_GAMMA_THERMALISATION_SCHEME ARTIS_GAMMA_FREQUENCYDEPENDENT
#endif

/* kpkt.cc:51 kpktdiffusion_timestep_fraction (a float in the reference) */
#define ARTIS_KPKTDIFFUSION_TIMESTEP_FRACTION 0.001f


#endif
